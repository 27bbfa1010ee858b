// nfl_capi.hip -- the extern "C" surface declared in include/nerf_fl_amd.h.
#include <hip/hip_runtime.h>
#include <string.h>

#include "../../include/nerf_fl_amd.h"
#include "nfl_plan.h"

extern "C" int nfl_launch_render_x3(const NflPlan*, const void*, const void*, const nfl_pass_args*, void*);
extern "C" int nfl_launch_render_x1(const NflPlan*, const void*, const void*, const nfl_pass_args*, void*);

extern "C" {

size_t nfl_plan_bytes(const nfl_field_desc*) { return sizeof(NflPlan); }

int nfl_plan_build(const nfl_field_desc* desc, int prec, void* h_plan, size_t bytes) {
    if (!h_plan) return NFL_EINVAL;
    if (bytes < sizeof(NflPlan)) return NFL_ESMALL;
    return nfl_plan_fill(desc, prec, static_cast<NflPlan*>(h_plan));
}

size_t nfl_packed_bytes(const nfl_field_desc* desc, int prec) {
    NflPlan p;
    if (nfl_plan_fill(desc, prec, &p) != NFL_OK) return 0;
    return (size_t)p.packed_bytes;
}

size_t nfl_param_count(const nfl_field_desc* d) {
    if (!d) return 0;
    const size_t cx = 6 * d->n_emb_xyz + 3, cd = 6 * d->n_emb_dir + 3, W = NFL_W, H = NFL_W / 2;
    const size_t na = d->encode_appearance ? d->n_a : 0;
    size_t n = (cx + 1) * W + 6 * (W + 1) * W + (W + cx + 1) * W;      // trunk
    n += (W + 1) * W + (W + cd + na + 1) * H + (W + 1) + (H + 1) * 3;  // final, dir, sigma, rgb
    if (d->encode_transient) n += (W + d->n_tau + 1) * H + 3 * (H + 1) * H + 5 * (H + 1);
    return n;
}

static size_t n_segments(int32_t n_rays, int32_t n_samples) {
    return (size_t)n_rays * (size_t)((n_samples + 31) / 32);
}
size_t nfl_act_stash_bytes(const nfl_field_desc* d, int32_t n_rays, int32_t n_samples, int32_t bwd_prec) {
    if (!d || n_rays < 0 || n_samples < 1 || bwd_prec < 0 || bwd_prec > NFL_PREC_F16W) return 0;
    if (d->n_emb_xyz < 1 || d->n_emb_xyz > NFL_MAX_EMB_XYZ) return 0;
    const int nkp = nfl_nkp_for(d->n_emb_xyz);
    const int mult = bwd_prec == NFL_PREC_F16X3 ? 2 : 1;
    // records (hi, with a three-product backward + lo) + tail pad for 2-k-step tile reads, then the relu-mask words
    return nfl_msk_offset(n_segments(n_rays, n_samples), nkp, mult) + n_segments(n_rays, n_samples) * NFL_MSK_WORDS * 256;
}
size_t nfl_grad_stash_bytes(const nfl_field_desc* d, int32_t n_rays, int32_t n_samples, int32_t bwd_prec) {
    if (!d || n_rays < 0 || n_samples < 1 || bwd_prec < 0 || bwd_prec > NFL_PREC_F16W) return 0;
    const int mult = bwd_prec == NFL_PREC_F16X3 ? 2 : 1;
    return (n_segments(n_rays, n_samples) + 1) * NFL_GRD_SLOTS * mult * 1024 + 4096;   // + one scratch record for padded segments
}
int nfl_bwd_plan_build(const nfl_field_desc* desc, int32_t rays_grad, int32_t bwd_prec, void* h_plan, size_t bytes) {
    if (!h_plan) return NFL_EINVAL;
    if (bytes < sizeof(NflPlan)) return NFL_ESMALL;
    return nfl_plan_fill_bwd(desc, rays_grad, bwd_prec, static_cast<NflPlan*>(h_plan));
}
size_t nfl_bwd_packed_bytes(const nfl_field_desc* desc, int32_t rays_grad, int32_t bwd_prec) {
    NflPlan p;
    if (nfl_plan_fill_bwd(desc, rays_grad, bwd_prec, &p) != NFL_OK) return 0;
    return (size_t)p.packed_bytes;
}

int nfl_render_pass(const void* h_plan, const void* d_plan, const void* d_packed,
                    const nfl_pass_args* a, void* stream) {
    const NflPlan* hp = static_cast<const NflPlan*>(h_plan);
    if (!hp || hp->magic != NFL_PLAN_MAGIC || !d_plan || !d_packed || !a) return NFL_EINVAL;
    if (a->n_rays < 0 || a->n_samples < 1 || (!a->d_rays && !a->h_cam && !a->d_cam)) return NFL_EINVAL;
    if (a->d_cam && a->d_act_stash) return NFL_EINVAL;
    if (a->h_cam && (a->d_act_stash || a->h_cam->width < 1 || a->h_cam->fx == 0.f || a->h_cam->fy == 0.f || a->h_cam->pix0 < 0))
        return NFL_EINVAL;
    if (!a->d_z && !a->d_lin) return NFL_EINVAL;
    if (a->perturb > 0.f && !a->d_z && !a->d_perturb_rand) return NFL_EINVAL;
    if (!a->sigma_only && hp->has_a && !a->d_a_emb) return NFL_EINVAL;
    if (hp->is_bwd) return NFL_EINVAL;
    if (a->d_loss_target && (!a->d_losses || !a->d_seed_rgb || a->loss_slot < 0 || a->loss_slot > 1 || a->sigma_only ||
                             a->test_extras)) return NFL_EINVAL;
    if (a->n_rays == 0) return NFL_OK;
    if (hp->prec == NFL_PREC_F16X3) return nfl_launch_render_x3(hp, d_plan, d_packed, a, stream);
    return nfl_launch_render_x1(hp, d_plan, d_packed, a, stream);
}

int nfl_field_forward(const void* h_plan, const void* d_plan, const void* d_packed, const float* d_x,
                      int32_t n_points, int32_t row_stride, int32_t sigma_only, int32_t output_transient,
                      float* d_out, void* stream) {
    const NflPlan* hp = static_cast<const NflPlan*>(h_plan);
    if (!hp || hp->magic != NFL_PLAN_MAGIC || hp->is_bwd || !d_plan || !d_packed || !d_x || !d_out) return NFL_EINVAL;
    const int cx = 6 * hp->n_emb_xyz + 3, cd = hp->ld[NFL_P_DIR] - NFL_W - hp->n_a;
    const int need = sigma_only ? cx : cx + cd + hp->n_a + ((output_transient && hp->has_t) ? hp->n_tau : 0);
    if (n_points < 0 || row_stride < need) return NFL_EINVAL;
    if (n_points == 0) return NFL_OK;
    nfl_pass_args a;
    memset(&a, 0, sizeof(a));
    a.d_embedded = d_x;
    a.n_points = n_points;
    a.embedded_stride = row_stride;
    a.n_rays = (n_points + 31) / 32;
    a.n_samples = 32;
    a.sigma_only = sigma_only ? 1 : 0;
    a.d_t_emb = (output_transient && hp->has_t && !sigma_only) ? d_x : nullptr;     // flag only
    a.d_a_emb = d_x;                                                                  // flag only
    a.d_rays = d_x;
    a.d_field_raw = d_out;
    if (hp->prec == NFL_PREC_F16X3) return nfl_launch_render_x3(hp, d_plan, d_packed, &a, stream);
    return nfl_launch_render_x1(hp, d_plan, d_packed, &a, stream);
}

int nfl_abi_version(void) { return NFL_ABI_VERSION; }

const char* nfl_version(void) { return "nerf_fl_amd 0.1 (gfx950, HIP; abi 8)"; }

const char* nfl_strerror(int code) {
    switch (code) {
        case NFL_OK: return "ok";
        case NFL_EINVAL: return "invalid argument or unsupported configuration";
        case NFL_ELAUNCH: return "HIP launch failed";
        case NFL_ENODEV: return "no usable gfx950 device";
        case NFL_ESMALL: return "buffer too small";
        default: return "unknown error";
    }
}

const char* nfl_render_kernel_name(int prec, int n_emb_xyz) {
    if (prec == NFL_PREC_F16X3) return n_emb_xyz > 10 ? "nfl_render_kernel<3, 1, 15, 0>" : "nfl_render_kernel<3, 1, 10, 0>";
    return n_emb_xyz > 10 ? "nfl_render_kernel<1, 1, 15, 0>" : "nfl_render_kernel<1, 1, 10, 0>";
}

}  // extern "C"
