// nfl_render_x1.hip -- instantiations of the fused render kernel, single-product fp16
// mode (fast; ~2^-11 relative per product).  One 32-sample segment per wave (two need a leaner register budget: TODO).
#include "nfl_render_impl.h"

extern "C" int nfl_launch_render_x1(const NflPlan* hp, const void* d_plan, const void* d_packed,
                                    const nfl_pass_args* args, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hp->n_emb_xyz <= 10) return nfl_launch_render<1, 1, 10>(hp, d_plan, d_packed, args, s);
    if (hp->n_emb_xyz <= 15) return nfl_launch_render<1, 1, 15>(hp, d_plan, d_packed, args, s);
    return NFL_EINVAL;
}
