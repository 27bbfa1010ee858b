// nfl_render_x3.hip -- instantiations of the fused render kernel, 3-product fp16 mode
// (fp32-class accuracy; default).  One 32-sample segment per wave.
#include "nfl_render_impl.h"

// The instantiations for up to 15 frequencies live in their own translation unit (nfl_render_x3w.hip) so that the two
// halves of the library's longest compile run in parallel.
extern "C" int nfl_launch_render_x3_wide(const NflPlan*, const void*, const void*, const nfl_pass_args*, void*);

extern "C" int nfl_launch_render_x3(const NflPlan* hp, const void* d_plan, const void* d_packed,
                                    const nfl_pass_args* args, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hp->n_emb_xyz <= 10) return nfl_launch_render<3, 1, 10>(hp, d_plan, d_packed, args, s);
    if (hp->n_emb_xyz <= 15) return nfl_launch_render_x3_wide(hp, d_plan, d_packed, args, stream);
    return NFL_EINVAL;
}

