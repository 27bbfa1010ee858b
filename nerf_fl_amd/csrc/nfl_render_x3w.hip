// nfl_render_x3w.hip -- the 3-product render kernel for encoders of 11..15 frequencies (6 k-steps of encoded position).
#include "nfl_render_impl.h"

extern "C" int nfl_launch_render_x3_wide(const NflPlan* hp, const void* d_plan, const void* d_packed,
                                         const nfl_pass_args* args, void* stream) {
    return nfl_launch_render<3, 1, 15>(hp, d_plan, d_packed, args, static_cast<hipStream_t>(stream));
}
