/*
 * nerf_fl_amd.h -- C ABI of the MI355X-native NeRF-W ray renderer.
 *
 * This is the drop-in boundary for ONE path of nmerty/nerf-fl: everything that
 * happens inside `render_rays` (reference models/rendering.py:49-289) and the
 * field it evaluates (reference models/nerf.py:6-32, 81-212).  The reference
 * has no FFI of its own (it is pure Python); these are the entry points a
 * ctypes stub in the reference's models/rendering.py would bind (see
 * INTEGRATION.md).  Conventions:
 *
 *   - plain pointers and sizes only; no torch / HIP types in the signatures
 *     (`stream` is a hipStream_t passed as void*),
 *   - every pointer named d_* is DEVICE memory owned by the caller,
 *   - nothing here allocates, frees or synchronises; all work is enqueued on
 *     `stream` (graph-capturable),
 *   - every function returns NFL_OK (0) or a negative NFL_E* code;
 *     nfl_strerror() names it.  The Python shim raises RuntimeError on != 0,
 *     which mirrors the reference's "plain Python exception" convention
 *     (SURVEY.md 8b).
 */
#ifndef NERF_FL_AMD_H
#define NERF_FL_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NFL_ABI_VERSION 9
#define NFL_GMAX_SLOTS 1024

enum {
    NFL_OK = 0,
    NFL_EINVAL = -1,      /* bad argument (null pointer, size, unsupported config) */
    NFL_ELAUNCH = -2,     /* hipLaunchKernel / hipMemcpyAsync failed             */
    NFL_ENODEV = -3,      /* no gfx950 device / kernel image not loadable         */
    NFL_ESMALL = -4       /* caller-provided buffer too small                     */
};

/* Arithmetic used inside the MLP (everything outside the matrix products --
 * positional encoding, activations, compositing, sampling -- is always fp32). */
#define NFL_STATUS_NONFINITE 1   /* a composited per-ray output of a render pass was NaN / inf                        */
#define NFL_STATUS_RANGE 2       /* a weight or an activation exceeded fp16's range (|x| > 65504) inside the fused MLP */

enum {
    NFL_PREC_F16X3 = 0,   /* fp16 MFMA, operands split hi+lo, 3 products: ~2^-21 relative, fp32-class (default) */
    NFL_PREC_F16   = 1,   /* fp16 MFMA, single product: ~2^-11 relative (fast mode)                               */
    NFL_PREC_F16W  = 2    /* BACKWARD only (bwd_prec): as NFL_PREC_F16, but the gradient chain multiplies by hi + lo weight
                             fragments (two products, the weights to fp32 class): no systematic training-curve offset   */
};

/* One field (reference class NeRF, models/nerf.py:80-151).  W=256, D=8,
 * skips=[4] are fixed, as in every configuration the reference ships. */
typedef struct nfl_field_desc {
    int32_t n_emb_xyz;          /* PosEmbedding freqs for xyz: 1..15 (opt.py:25 default 10; test_phototourism.ipynb 15)  */
    int32_t n_emb_dir;          /* PosEmbedding freqs for dir: 1..4 (opt.py:27 default 4)                                */
    int32_t encode_appearance;  /* NeRF-A head: dir layer sees n_a extra inputs (nerf.py:115,134)             */
    int32_t n_a;                /* 1..48 (opt.py --N_a, default 48)                                           */
    int32_t encode_transient;   /* NeRF-U head (nerf.py:141-151)                                              */
    int32_t n_tau;              /* 1..16 (opt.py --N_tau, default 16)                                         */
    float   beta_min;           /* added after compositing (rendering.py:185)                                 */
    int32_t reserved;
} nfl_field_desc;

/* Device pointers to the fp32 parameters of one field, `nn.Linear` layout
 * (weight = (out,in) row-major, bias = (out)); index with NFL_P_*.
 * Names are the reference's state_dict keys. */
enum {
    NFL_P_XYZ1 = 0,  /* xyz_encoding_1.0 ... xyz_encoding_8.0 = NFL_P_XYZ1 + i */
    NFL_P_FINAL = 8, /* xyz_encoding_final      */
    NFL_P_DIR = 9,   /* dir_encoding.0          */
    NFL_P_SIGMA = 10,/* static_sigma.0          */
    NFL_P_RGB = 11,  /* static_rgb.0            */
    NFL_P_T0 = 12,   /* transient_encoding.0, .2, .4, .6 = NFL_P_T0 + j */
    NFL_P_TSIGMA = 16,
    NFL_P_TRGB = 17,
    NFL_P_TBETA = 18,
    NFL_NUM_LAYERS = 19
};
typedef struct nfl_field_params {
    const float* weight[NFL_NUM_LAYERS];
    const float* bias[NFL_NUM_LAYERS];
} nfl_field_params;

/* ---- plan: static tables describing the packed weight stream ------------ */

/* Bytes of the (host-built, then caller-uploaded) plan for this field. */
size_t nfl_plan_bytes(const nfl_field_desc* desc);
/* Fill `h_plan` (host memory, nfl_plan_bytes) for `prec`.  The caller copies it
 * verbatim to device memory and passes that pointer as d_plan below. */
int nfl_plan_build(const nfl_field_desc* desc, int prec, void* h_plan, size_t bytes);
/* Bytes of the packed weight stream (fp16 MFMA fragments + fp32 bias table). */
size_t nfl_packed_bytes(const nfl_field_desc* desc, int prec);
/* Number of fp32 parameters of the field (sum of numel), for gradient arenas. */
size_t nfl_param_count(const nfl_field_desc* desc);

/* Re-pack the current fp32 parameters into MFMA fragment order (run after every
 * optimizer step; ~2.4 MB read, <=4.8 MB written). h_plan is the same table in
 * HOST memory (used only to size the launch).  d_status (device int32, may be NULL): NFL_STATUS_RANGE is OR-ed in
 * when a weight does not fit fp16's range. */
int nfl_pack_field(const void* h_plan, const void* d_plan, const nfl_field_params* params,
                   void* d_packed, size_t packed_bytes, int32_t* d_status, void* stream);
/* xyz_encoding_final is linear and feeds dir_encoding.0 / transient_encoding.0 only, so the packed streams carry no tiles
 * for it: it is folded into the first 256 input columns of those two layers,
 *   W' = [W[:, :256] W_fin | W[:, 256:]],   b' = b + W[:, :256] b_fin.
 * n_side = side-input columns of dir_encoding.0 = (6 n_emb_dir + 3) + n_a (0 without the appearance input).
 * d_wdir_c (128, 256 + n_side) and d_wt0_c (128, 256 + n_tau; with has_t) must hold COPIES of dir_encoding.0.weight /
 * transient_encoding.0.weight on entry (their side columns are kept); d_bdir_c / d_bt0_c (128) are written.  Pack with a
 * nfl_field_params whose weight / bias entries NFL_P_DIR (and NFL_P_T0) point at these folded tensors; every other
 * entry point (nfl_mlp_wgrad's `params`) takes the ORIGINAL parameters.  One launch, fp32 (v_mfma_f32_32x32x2_f32). */
int nfl_compose_forward(const nfl_field_params* params, int32_t has_t, int32_t n_side, int32_t n_tau,
                        float* d_wdir_c, float* d_bdir_c, float* d_wt0_c, float* d_bt0_c, void* stream);
/* Several streams in one launch (a training step re-packs the forward and the dgrad stream of both fields after every
 * optimizer update).  Same argument meaning and checks as nfl_pack_field, per job. */
#define NFL_PACK_MAX_JOBS 4
typedef struct nfl_pack_job {
    const void* h_plan;
    const void* d_plan;
    const nfl_field_params* params;
    void* d_packed;
    size_t packed_bytes;
    int32_t* d_status;          /* may be NULL */
} nfl_pack_job;
int nfl_pack_fields(int32_t n_jobs, const nfl_pack_job* jobs, void* stream);

/* ---- one rendering pass (reference: inference(), rendering.py:83-226) --- */

/* A pinhole camera from which a pass can generate its rays in the prologue instead of reading a ray matrix (reference
 * datasets/ray_utils.py:5-55, get_ray_directions + get_rays; same arithmetic as nfl_gen_rays): ray r of the pass is
 * pixel pix0 + r in row-major order of a frame `width` pixels wide; direction [(i - cx) / fx, -(j - cy) / fy, -1] (no
 * half-pixel) rotated by c2w[:, :3] and normalised, origin c2w[:, 3], bounds near / far. */
typedef struct nfl_camera {
    float   c2w[12];            /* 3 x 4 row-major */
    float   fx, fy, cx, cy;
    int32_t width, reserved;
    int64_t pix0;
    float   near, far;
} nfl_camera;

typedef struct nfl_pass_args {
    /* geometry */
    const float* d_rays;        /* (R,8): o(3) d(3) near far   (rendering.py:231-233); may be NULL when h_cam / d_cam is set */
    const nfl_camera* h_cam;    /* HOST pointer or NULL: generate the rays of this pass from the camera (copied at launch;
                                   inference only: not with d_act_stash)                                             */
    const float* d_view_dir;    /* (R,3) or NULL -> use rays_d (rendering.py:236-238)  */
    int32_t n_rays;             /* R                                                     */
    int32_t n_samples;          /* samples per ray in THIS pass: N_samples (coarse) or N_samples+N_importance (fine) */
    /* depths: either given (fine pass: sorted output of nfl_sample_pdf) ...      */
    const float* d_z;           /* (R,n_samples) or NULL                                 */
    /* ... or generated in-kernel (coarse pass, rendering.py:243-259)              */
    const float* d_lin;         /* (n_samples) = linspace(0,1,n_samples); required when d_z == NULL */
    const float* d_perturb_rand;/* (R,n_samples) U[0,1) or NULL when perturb == 0         */
    float   perturb;
    int32_t use_disp;
    float*  d_z_out;            /* (R,n_samples) or NULL: the depths actually used       */
    /* density noise (rendering.py:151-152; ignored when the transient head is on) */
    const float* d_noise;       /* (R,n_samples) N(0,1) or NULL                          */
    float   noise_std;
    /* latent codes, already looked up per ray (rendering.py:276-286)              */
    const float* d_a_emb;       /* (R,n_a)  or NULL                                      */
    const float* d_t_emb;       /* (R,n_tau) or NULL -> transient head off for this pass */
    /* mode */
    int32_t sigma_only;         /* coarse pass at test_time (rendering.py:103-111,169)   */
    int32_t white_back;
    int32_t test_extras;        /* test_time on a transient pass: rgb/depth_fine_{static,transient} */
    int32_t stash_split;        /* with d_act_stash: 1 = write split (hi + lo) activation records for a backward in
                                   NFL_PREC_F16X3 (size the stash with nfl_act_stash_bytes(..., NFL_PREC_F16X3)); 0 = hi only */
    /* outputs (any may be NULL = not wanted) */
    float* d_weights;           /* (R,n_samples)                                         */
    float* d_opacity;           /* (R)                                                   */
    float* d_rgb;               /* (R,3)  rgb_coarse / rgb_fine                          */
    float* d_depth;             /* (R)                                                   */
    float* d_transient_sigmas;  /* (R,n_samples)                                         */
    float* d_beta;              /* (R)                                                   */
    float* d_rgb_static;        /* (R,3)  _rgb_fine_static                               */
    float* d_rgb_transient;     /* (R,3)  _rgb_fine_transient                            */
    float* d_rgb_static_only;   /* (R,3)  rgb_fine_static   (test_extras)                */
    float* d_depth_static_only; /* (R)    depth_fine_static                              */
    float* d_rgb_transient_only;/* (R,3)  rgb_fine_transient                             */
    float* d_depth_transient_only;/* (R)  depth_fine_transient                           */
    float* d_field_raw;         /* (R*n_samples,9) per-sample field outputs [rgb,sigma,rgb_t,sigma_t,beta] or NULL;
                                   required (with d_act_stash) when a backward will follow                      */
    char*  d_act_stash;         /* nfl_act_stash_bytes(): fp16 layer inputs of every sample in MFMA fragment order, then
                                   the relu-mask words; consumed by nfl_mlp_dgrad / nfl_mlp_wgrad; NULL for inference */
    /* BARF coarse-to-fine encoding (reference BarfPosEmbedding, nerf.py:35-77): per-frequency weights,
       computed by the caller exactly as barf_weight(freq, epoch) does; NULL = plain PosEmbedding     */
    const float* d_pe_w_xyz;    /* (n_emb_xyz) or NULL                                                   */
    const float* d_pe_w_dir;    /* (n_emb_dir) or NULL                                                   */
    /* NerfWLoss fused into the per-ray epilogue of a TRAINING pass (reference losses.py:35-50): with
       d_loss_target set, every completed ray adds its share of the loss terms to d_losses[4] = {c_l, f_l, b_l, s_l}
       (float atomics, one flush per workgroup; the caller zeroes the array once per step) -- loss_slot 0: this is the
       coarse pass (c_l); 1: the fine pass (f_l, and with the transient head b_l and s_l) -- and writes the backward
       seeds d loss / d rgb (and d loss / d beta) of the ray, which nfl_composite_backward takes as g_rgb / g_beta.
       The caller then needs none of the per-sample outputs (weights, transient_sigmas) for the loss. */
    const float* d_loss_target; /* (R,3) target colours, or NULL = no fused loss                                   */
    float*  d_losses;           /* (4)                                                                              */
    float*  d_seed_rgb;         /* out (R,3)                                                                        */
    float*  d_seed_beta;        /* out (R); fine pass with the transient head only                                  */
    float   loss_coef, lambda_u;/* losses.py:36: coef = 1, lambda_u = 0.01                                          */
    int32_t loss_slot, reserved2;
    /* optional status word (device, int32, never cleared by the library): NFL_STATUS_* bits are OR-ed in.  The MLP
       multiplies fp16 operands: an activation or a weight beyond fp16's range (|x| > 65504), which the fp32 reference
       would carry, cannot be represented here (hi = inf, lo = -inf; the matrix cores then produce NaNs that the next
       relu turns into zeros, so the outputs may even look plausible) -- the ABI's range limit.  The
       kernels track the largest fp16 operand they form and report NFL_STATUS_RANGE; this word is how a caller learns
       of it without scanning anything. */
    int32_t* d_status;
    /* used by nfl_field_forward only (leave NULL / 0 otherwise) */
    const float* d_embedded;
    int32_t n_points, embedded_stride;
    /* the camera in DEVICE memory (same struct as h_cam; takes precedence over it): the launch then carries only the
       pointer, so a pass captured in a HIP graph renders whatever camera the buffer holds at replay time (h_cam is copied
       into the launch's arguments and would be frozen by the capture).  Must not change while the pass runs. */
    const nfl_camera* d_cam;
} nfl_pass_args;

/* Evaluate the field on every sample of every ray and alpha-composite on the
 * fly (one fused kernel; per-sample activations never reach HBM). */
int nfl_render_pass(const void* h_plan, const void* d_plan, const void* d_packed,
                    const nfl_pass_args* args, void* stream);

/* ---- the field alone (reference NeRF.forward, models/nerf.py:153-212) ---------
 * d_x (n_points, row_stride) fp32 = [encoded xyz | encoded dir (+ appearance) | tau], exactly the
 * matrix the reference module takes; d_out (n_points, 9) = [rgb(3), sigma, rgb_t(3), sigma_t, beta]
 * (columns the mode does not compute are written as 0).  Same fused MFMA kernel as the
 * renderer, with the encoding/compositing stages compiled out. */
int nfl_field_forward(const void* h_plan, const void* d_plan, const void* d_packed,
                      const float* d_x, int32_t n_points, int32_t row_stride,
                      int32_t sigma_only, int32_t output_transient, float* d_out, void* stream);
/* reference PosEmbedding / BarfPosEmbedding.forward (models/nerf.py:19-32, 61-77): d_x (n,3) ->
 * d_out (n, 6*n_freqs+3); d_w (n_freqs) BARF weights or NULL */
int nfl_posenc(const float* d_x, int32_t n, int32_t n_freqs, const float* d_w, float* d_out, void* stream);
/* reference datasets/ray_utils.py:5-55 (get_ray_directions + get_rays) for `count` pixels of a frame of width `width`,
 * starting at row-major pixel index `start`: camera direction [(i-cx)/fx, -(j-cy)/fy, -1] (no half-pixel), rotated by
 * c2w[:, :3] and normalised; origin c2w[:, 3].  h_c2w: 12 floats, 3x4 row-major, on the HOST.  d_rays (count, 8) =
 * [origin, direction, near, far], the matrix nfl_render_pass takes. */
int nfl_gen_rays(const float* h_c2w, float fx, float fy, float cx, float cy, int32_t width, int64_t start,
                 int32_t count, float near, float far, float* d_rays, void* stream);

/* ---- backward (training) ---------------------------------------------------
 * The reference gets its gradients from autograd replaying ~100 ATen kernels per
 * point chunk over saved (chunk,256) activations.  Here:
 *   forward (d_act_stash, d_field_raw set)  ->  nfl_composite_backward  ->  nfl_mlp_dgrad
 *   ->  nfl_mlp_wgrad.  All gradients are returned in fp32.  The MLP part of the backward is
 *   mixed precision: fp16 MFMA with fp32 accumulation on activations stashed in fp16 and on
 *   gradients multiplied by a per-pass power-of-two LOSS SCALE, chosen on the device from
 *   max|head gradient| (d_gmax, written by nfl_composite_backward) so that fp16's range is
 *   used whatever the loss magnitude; the scale is divided out before anything is returned.
 *
 *   bwd_prec selects the arithmetic of that MLP part (profiles/r03_psnr_backward_attribution.txt):
 *     NFL_PREC_F16   (default) one fp16 product everywhere: gradients within a few 1e-3 of fp32 autograd per step, fastest;
 *                    the fp16-rounded transposed weights of the chain leave a small systematic offset in long training
 *                    curves (-0.4 .. -1 % of the late training loss on the NeRF-W parity scene; validation PSNR unaffected);
 *     NFL_PREC_F16W  the chain delta_{l-1} = W_l^T delta_l reads hi + lo weight fragments (two products): that offset is
 *                    gone (curve within the reference's own run-to-run scatter); stashes and weight gradients as F16;
 *     NFL_PREC_F16X3 the forward's split-operand arithmetic throughout (hi + lo weights, activations and gradients, three
 *                    products, weight gradients from hi + lo records in one pass): fp32-class gradients, the precision class of the
 *                    reference's autograd; the stashes then hold a second, residual record per segment (twice the bytes),
 *                    written by a forward pass with nfl_pass_args::stash_split = 1.
 *   One value must be used for the stash sizes, the forward pass, the dgrad plan / stream and nfl_mlp_wgrad of a step. */
size_t nfl_act_stash_bytes(const nfl_field_desc* desc, int32_t n_rays, int32_t n_samples, int32_t bwd_prec);
size_t nfl_grad_stash_bytes(const nfl_field_desc* desc, int32_t n_rays, int32_t n_samples, int32_t bwd_prec);
/* dgrad plan / packed stream (transposed weights, fp16); same calling pattern as
 * nfl_plan_build / nfl_pack_field (pack with nfl_pack_field using these plans). */
/* rays_grad != 0: the stream also carries the tiles needed for the gradient w.r.t. the rays
 * (learnable poses, reference models/poses.py + train.py:86-98). */
int    nfl_bwd_plan_build(const nfl_field_desc* desc, int32_t rays_grad, int32_t bwd_prec, void* h_plan, size_t bytes);
size_t nfl_bwd_packed_bytes(const nfl_field_desc* desc, int32_t rays_grad, int32_t bwd_prec);

typedef struct nfl_compbwd_args {
    const float* d_field_raw;       /* (R*N,9) from the forward pass                       */
    const float* d_z;               /* (R,N) depths the forward pass used                  */
    const float* d_noise;           /* (R,N) or NULL, as given to the forward pass         */
    float   noise_std;
    int32_t n_rays, n_samples;
    int32_t use_transient;          /* the forward pass evaluated the transient head       */
    int32_t white_back;
    int32_t reserved;
    /* gradients of the pass outputs; NULL = zero */
    const float* g_weights;         /* (R,N)  */
    const float* g_opacity;         /* (R)    */
    const float* g_rgb;             /* (R,3)  rgb_coarse / rgb_fine                        */
    const float* g_depth;           /* (R)    */
    const float* g_transient_sigmas;/* (R,N)  */
    const float* g_beta;            /* (R)    */
    const float* g_rgb_static;      /* (R,3)  _rgb_fine_static                             */
    const float* g_rgb_transient;   /* (R,3)  _rgb_fine_transient                          */
    float g_tsig_const;             /* added to every element of g_transient_sigmas (s_l's constant gradient coef lambda_u / (R N): no (R,N) array needed) */
    int32_t reserved3;
    const float* d_go;              /* device scalar multiplying every gradient above (the upstream gradient of a fused loss) or NULL = 1 */
    float* d_head_grads;            /* out (R*N,9): d/d pre-activation [rgb,sigma,rgb_t,sigma_t,beta] */
    float* d_gmax;                  /* out (NFL_GMAX_SLOTS = 1024 floats): partial maxima of |head gradient| of this pass (zeroed by the call; the consumers take the max) */
} nfl_compbwd_args;
int nfl_composite_backward(const nfl_compbwd_args* args, void* stream);

typedef struct nfl_dgrad_args {
    const float* d_head_grads;      /* (R*N,9) from nfl_composite_backward                 */
    const char*  d_act_stash;       /* from the forward pass                               */
    char*        d_grad_stash;      /* out, nfl_grad_stash_bytes()                         */
    int32_t n_rays, n_samples;
    int32_t use_transient;
    int32_t dir_is_data;            /* with d_g_rays: the forward pass was given a separate d_view_dir, so the direction encoding
                                       does not depend on the rays (rendering.py:236-238): its gradient is left out of d_g_rays */
    float* d_g_a_emb;               /* (R,n_a)  accumulated into (zero it first) or NULL   */
    float* d_g_t_emb;               /* (R,n_tau) accumulated into (zero it first) or NULL  */
    const int64_t* d_latent_row;    /* (R) or NULL.  When set, d_g_a_emb / d_g_t_emb are the gradients of the latent TABLES
                                       (N_vocab, n_a) / (N_vocab, n_tau) and ray r accumulates into row d_latent_row[r] (= ts[r]):
                                       the scatter-add of nn.Embedding's backward (rendering.py:276-286) happens in this kernel  */
    /* gradient w.r.t. the rays (needs a plan built with rays_grad = 1) */
    float* d_g_rays;                /* (R,8) accumulated into: columns 0..2 origin, 3..5 direction; 6,7 untouched; or NULL */
    const float* d_rays;            /* (R,8) as given to the forward pass                  */
    const float* d_z;               /* (R,N) depths the forward pass used                  */
    const float* d_pe_w_xyz;        /* as given to the forward pass (NULL = ones)          */
    const float* d_pe_w_dir;
    const float* d_gmax;            /* (1024) from nfl_composite_backward: fixes the loss scale of this pass */
    uint32_t rounding_seed;         /* NFL_PREC_F16 / NFL_PREC_F16W round their fp16 gradients STOCHASTICALLY (unbiased; see
                                       nfl_mlp_dgrad below).  The draws are a function of (this seed, the work-item, *d_gmax):
                                       the same seed on the same data reproduces them, another seed gives an independent set */
} nfl_dgrad_args;
/* The gradient chain delta_{l-1} = relu'(h_{l-1}) W_l^T delta_l through the field, from the head gradients down to the
 * latent codes (and the rays), every delta_l left in d_grad_stash for nfl_mlp_wgrad.  Arithmetic by the plan's bwd_prec.
 * Rounding (NFL_PREC_F16 / NFL_PREC_F16W): every fp32 -> fp16 conversion of a gradient is STOCHASTIC (v_cvt_sr_f16_f32: up with
 * probability = the discarded fraction), so that its error is zero-mean and independent between samples and averages out of
 * the weight-gradient sums; NFL_PREC_F16's packed transposed weights are rounded the same way by nfl_pack_field (the draw a
 * hash of the weight's bits and position: deterministic, redrawn when the weight changes).  NFL_PREC_F16X3 carries hi + lo
 * parts instead and rounds to nearest.  (The reference differentiates in fp32: train.py:158-174.) */
int nfl_mlp_dgrad(const void* h_bwd_plan, const void* d_bwd_plan, const void* d_bwd_packed,
                  const nfl_dgrad_args* args, void* stream);

/* fp32 gradient tensors in nn.Linear layout, WRITTEN by nfl_mlp_wgrad (it zeroes them, accumulates
 * with atomics, then divides the loss scale out); entries may be NULL (layer absent / gradient
 * not wanted). */
typedef struct nfl_field_grads {
    float* weight[NFL_NUM_LAYERS];
    float* bias[NFL_NUM_LAYERS];
} nfl_field_grads;
/* wgrad plan: the list of per-layer streaming GEMM jobs (host-built once per field and
 * transient on/off; the caller uploads it verbatim like the other plans). */
size_t nfl_wgrad_plan_bytes(void);
int    nfl_wgrad_plan_build(const nfl_field_desc* desc, int32_t use_transient, void* h_plan, size_t bytes);
/* d_gmax: the same 1024 floats the dgrad of this pass was given (the stashed gradients carry its loss scale).
 * params / d_scratch: xyz_encoding_final is linear, so neither its output nor the gradient w.r.t. its output is ever
 * stashed; the gradients of xyz_encoding_final and of the first 256 input columns of dir_encoding.0 /
 * transient_encoding.0 are composed, in fp32, from G = sum_s delta_dirh (x) h8 (in d_scratch) and the CURRENT fp32 weights of those layers (`params`: the weights the
 * forward pass ran with; weight[NFL_P_FINAL], bias[NFL_P_FINAL], weight[NFL_P_DIR] and, with the transient head,
 * weight[NFL_P_T0] are read).  grads->bias[NFL_P_DIR] (and [NFL_P_T0]) must be given when any composed gradient is.
 * bwd_prec: the value the stashes were sized and written with.  NFL_PREC_F16 / NFL_PREC_F16W: dW = sum_s d_hi (x) h_hi,
 * one streaming pass; NFL_PREC_F16X3: the stashes hold residual records too and every accumulator gets
 * d_hi (x) h_hi + d_lo (x) h_hi + d_hi (x) h_lo in one pass over both (db = sum_s (d_hi + d_lo)).
 * d_scratch (nfl_wgrad_scratch_bytes() = 68 MB, overwritten; reusable by the next call on the same stream) holds G and the
 * PARTIAL SUMS of the streaming kernel's workgroups: every workgroup stores its accumulators there and a reduction launch adds
 * them up in a fixed order, divides by the loss scale and writes the gradient tensors -- the weight and bias gradients are
 * therefore bit-reproducible from run to run (no atomics), and elements no job owns (heads the call leaves out) are zero.
 * The gradient tensors may be views of one caller-owned flat buffer (nerf_fl_amd.parallel.GradArena): they are written in
 * place, so a collective can run on that buffer right after this call. */
size_t nfl_wgrad_scratch_bytes(void);
int nfl_mlp_wgrad(const void* h_wplan, const void* d_wplan, const char* d_act_stash, const char* d_grad_stash,
                  const float* d_gmax, int32_t n_rays, int32_t n_samples, int32_t bwd_prec,
                  const nfl_field_params* params, float* d_scratch, const nfl_field_grads* grads, void* stream);

/* ---- optimiser step (reference utils/__init__.py:30-32: torch.optim.Adam(lr, eps=1e-8), no weight decay, no
 * amsgrad) over up to NFL_ADAM_MAX_TENSORS fp32 tensors in one launch.  `step` is the 1-based count of this update
 * (bias corrections 1 - beta^step); a tensor whose grad pointer is NULL is left untouched. */
#define NFL_ADAM_MAX_TENSORS 64
typedef struct nfl_adam_tensors {
    float*       param[NFL_ADAM_MAX_TENSORS];
    const float* grad[NFL_ADAM_MAX_TENSORS];
    float*       exp_avg[NFL_ADAM_MAX_TENSORS];
    float*       exp_avg_sq[NFL_ADAM_MAX_TENSORS];
    int32_t      numel[NFL_ADAM_MAX_TENSORS];
} nfl_adam_tensors;
int nfl_adam_step(const nfl_adam_tensors* tensors, int32_t n_tensors, float lr, float beta1, float beta2, float eps,
                  int32_t step, void* stream);
/* The same update with its scalars in DEVICE memory, for launches captured in a HIP graph (a captured launch freezes
 * by-value arguments): d_hyper = float[4] {lr, beta1, beta2, eps}; *d_step = number of updates already applied to
 * these tensors (the kernel uses *d_step + 1).  bump != 0: a second, one-thread launch then increments *d_step; pass 0
 * for all but the last call when more than NFL_ADAM_MAX_TENSORS tensors share one counter. */
int nfl_adam_step_dev(const nfl_adam_tensors* tensors, int32_t n_tensors, const float* d_hyper, int32_t* d_step,
                      int32_t bump, void* stream);

/* ---- NerfWLoss (reference losses.py:35-50) on the renderer's outputs, forward and backward in one launch each.
 * Terms (d_losses[4], zeroed by nfl_loss_forward): c_l, f_l, b_l, s_l; f_l uses beta when d_beta != NULL
 * (then d_rgb_fine is required; b_l and, with d_transient_sigmas, s_l are produced too).
 * nfl_loss_backward writes the gradients of sum_k *d_grad_loss[k] * loss_k (a NULL entry counts as 0) w.r.t. rgb_coarse,
 * rgb_fine, beta and transient_sigmas (the last one is the constant coef*lambda_u/(R*N) per element). */
typedef struct nfl_loss_args {
    const float* d_rgb_coarse;        /* (R,3)                                  */
    const float* d_rgb_fine;          /* (R,3) or NULL (coarse-only rendering)  */
    const float* d_beta;              /* (R) or NULL                            */
    const float* d_transient_sigmas;  /* (R,N) or NULL                          */
    const float* d_target;            /* (R,3)                                  */
    int32_t n_rays, n_samples;        /* n_samples: columns of transient_sigmas */
    float   coef, lambda_u;           /* losses.py:36: coef = 1, lambda_u = 0.01 */
    float*  d_losses;                 /* out (4)                                */
    const float* d_grad_loss[4];      /* device scalars: d total / d {c_l, f_l, b_l, s_l}; NULL = 0 */
    float*  d_g_rgb_coarse;           /* out (R,3)                              */
    float*  d_g_rgb_fine;             /* out (R,3)                              */
    float*  d_g_beta;                 /* out (R)                                */
    float*  d_g_transient_sigmas;     /* out (R,N) or NULL                      */
} nfl_loss_args;
int nfl_loss_forward(const nfl_loss_args* args, void* stream);
int nfl_loss_backward(const nfl_loss_args* args, void* stream);

/* ---- hierarchical sampling (reference sample_pdf, rendering.py:7-46, plus the
 * concat + sort of rendering.py:267-272) -------------------------------------
 * d_z_coarse (R,S), d_weights_coarse (R,S); d_u (R,I) or NULL with d_u_row (I)
 * = linspace(0,1,I) shared by all rays (det).  Outputs: d_z_fine (R,S+I) sorted
 * ascending; d_samples (R,I) unsorted draws or NULL. */
int nfl_sample_pdf(const float* d_z_coarse, const float* d_weights_coarse,
                   const float* d_u, const float* d_u_row,
                   int32_t n_rays, int32_t n_samples, int32_t n_importance,
                   float* d_z_fine, float* d_samples, void* stream);

/* ---- misc ---------------------------------------------------------------- */
int         nfl_abi_version(void);
const char* nfl_version(void);        /* "nerf_fl_amd <x.y> gfx950 ..." */
const char* nfl_strerror(int code);
/* name of the dominant kernel of nfl_render_pass for (prec, n_emb_xyz), as it
 * appears in rocprofv3 kernel traces (used by bench.py to pair profiles) */
const char* nfl_render_kernel_name(int prec, int n_emb_xyz);

#ifdef __cplusplus
}
#endif
#endif /* NERF_FL_AMD_H */
