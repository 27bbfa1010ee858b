// nfl_compbwd.hip -- backward of the volume-rendering compositing, one wave per ray.
//
// Hand-written gradient of reference models/rendering.py:141-226 (what autograd
// replays there as ~40 small kernels plus a cumprod backward).  For every sample it
// turns the gradients of the per-ray outputs into gradients of the field's
// PRE-activation head outputs, which the dgrad kernel then pushes through the MLP.
//
// With g_i the total gradient reaching weight w_i = alpha_i T_i:
//   dL/dc_i      = w_i dL/drgb
//   dL/dsigma'_i = delta_i ( T_{i+1} g_i - sum_{j>i} w_j g_j )
// and, with the transient head on, the same suffix term sum_{j>i}(w_s g_s + w_t g_t + w g)_j
// is shared by sigma_s and sigma_t because both enter every later transmittance.
// The transmittance is an inclusive wavefront product scan, the suffix sums a reversed
// wavefront sum scan; rays longer than 64 samples are walked in 64-sample blocks with
// scalar carries (forward for T, backward for the suffix).
//
// HBM-bound: reads 36+8(+4) B, writes 36 B per sample.  Also leaves max|head gradient| of the pass in d_gmax:
// the fp16 MLP backward (nfl_dgrad / nfl_wgrad) derives its power-of-two loss scale from it.
#include <hip/hip_runtime.h>

#include "../../include/nerf_fl_amd.h"
#include "nfl_plan.h"

#define NFL_CB_MAXN 1024

__device__ __forceinline__ float cb_scan_mul(float v, int lane) {      // inclusive product scan
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float n = __shfl_up(v, d);
        if (lane >= d) v *= n;
    }
    return v;
}
__device__ __forceinline__ float cb_rscan_add(float v, int lane) {     // inclusive suffix-sum scan
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float n = __shfl_down(v, d);
        if (lane + d < 64) v += n;
    }
    return v;
}

__global__ __launch_bounds__(256) void nfl_compbwd_kernel(nfl_compbwd_args a) {
    __shared__ float T_lds[4][NFL_CB_MAXN];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ray = blockIdx.x * 4 + wave;
    if (ray >= a.n_rays) return;
    const int N = a.n_samples;
    float* Ts = T_lds[wave];
    const float* fr = a.d_field_raw + (size_t)ray * N * 9;
    const float* zr = a.d_z + (size_t)ray * N;
    const bool tr = a.use_transient != 0;
    const float ns = a.noise_std;

    // per-ray output gradients (uniform)
    const float go = a.d_go ? *a.d_go : 1.f;          // upstream gradient of a fused loss (header: d_go)
    auto ld = [&](const float* p, int k) { return p ? go * p[k] : 0.f; };
    const float gW = ld(a.g_opacity, ray), gD = ld(a.g_depth, ray), gB = ld(a.g_beta, ray);
    const float gts = go * a.g_tsig_const;
    float gCs[3], gCt[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float g = ld(a.g_rgb, ray * 3 + k);
        gCs[k] = g + (tr ? ld(a.g_rgb_static, ray * 3 + k) : 0.f);
        gCt[k] = g + ld(a.g_rgb_transient, ray * 3 + k);
    }
    const float gwhite = a.white_back ? gCs[0] + gCs[1] + gCs[2] : 0.f;

    // ---- pass 1: transmittance T_i (exclusive product of 1 - alpha), block by block
    float carry = 1.f;
    for (int i0 = 0; i0 < N; i0 += 64) {
        const int i = i0 + lane;
        float om = 1.f;
        if (i < N) {
            const float dl = i + 1 < N ? zr[i + 1] - zr[i] : 1e2f;
            const float sg = fr[i * 9 + 3];
            float al;
            if (tr) {
                al = 1.f - expf(-dl * (sg + fr[i * 9 + 7]));
            } else {
                const float nz = a.d_noise ? a.d_noise[(size_t)ray * N + i] * ns : 0.f;
                al = 1.f - expf(-dl * fmaxf(sg + nz, 0.f));
            }
            om = 1.f - al;
        }
        const float inc = cb_scan_mul(om, lane);
        float exc = __shfl_up(inc, 1);
        if (lane == 0) exc = 1.f;
        if (i < N) Ts[i] = carry * exc;
        carry *= __shfl(inc, 63);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- pass 2: suffix sums and head gradients, last block first
    float gmax = 0.f;        // max |head gradient| seen by this lane (loss scale of the fp16 MLP backward)
    float suffix = 0.f;      // sum of H_j over all samples after the current block
    const int nblk = (N + 63) / 64;
    for (int b = nblk - 1; b >= 0; --b) {
        const int i = b * 64 + lane;
        const bool ok = i < N;
        const int ii = ok ? i : N - 1;
        const float dl = ii + 1 < N ? zr[ii + 1] - zr[ii] : 1e2f;
        const float z = zr[ii], T = Ts[ii];
        const float* f = fr + ii * 9;
        const float cr = f[0], cg = f[1], cb = f[2], sg = f[3];
        const float gw = a.g_weights ? go * a.g_weights[(size_t)ray * N + ii] : 0.f;
        const float g = gw + gW + gD * z - gwhite;
        float out[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) out[k] = 0.f;
        float H;
        if (tr) {
            const float trr = f[4], tgg = f[5], tbb = f[6], sgt = f[7], bt = f[8];
            const float es = expf(-dl * sg), et = expf(-dl * sgt), ec = expf(-dl * (sg + sgt));
            const float a_s = 1.f - es, a_t = 1.f - et, al = 1.f - ec;
            const float gs = gCs[0] * cr + gCs[1] * cg + gCs[2] * cb;
            const float gt = gCt[0] * trr + gCt[1] * tgg + gCt[2] * tbb + gB * bt;
            const float ws = a_s * T, wt = a_t * T, w = al * T;
            H = ok ? ws * gs + wt * gt + w * g : 0.f;
            const float inc = cb_rscan_add(H, lane);
            const float after = inc - H + suffix;                 // sum over j > i
            const float common = ec * T * g - after;              // (1-alpha) T g - suffix
            const float dss = dl * (es * T * gs + common);
            const float dst = dl * (et * T * gt + common) + gts
                              + (a.g_transient_sigmas ? go * a.g_transient_sigmas[(size_t)ray * N + ii] : 0.f);
            out[0] = ws * gCs[0] * cr * (1.f - cr);
            out[1] = ws * gCs[1] * cg * (1.f - cg);
            out[2] = ws * gCs[2] * cb * (1.f - cb);
            out[3] = dss * (1.f - expf(-sg));                     // softplus'(x) = 1 - exp(-softplus(x))
            out[4] = wt * gCt[0] * trr * (1.f - trr);
            out[5] = wt * gCt[1] * tgg * (1.f - tgg);
            out[6] = wt * gCt[2] * tbb * (1.f - tbb);
            out[7] = dst * (1.f - expf(-sgt));
            out[8] = wt * gB * (1.f - expf(-bt));
            suffix += __shfl(inc, 0);
        } else {
            const float nz = a.d_noise ? a.d_noise[(size_t)ray * N + ii] * ns : 0.f;
            const float pre = sg + nz;
            const float e = expf(-dl * fmaxf(pre, 0.f));
            const float w = (1.f - e) * T;
            const float gg = g + gCs[0] * cr + gCs[1] * cg + gCs[2] * cb;
            H = ok ? w * gg : 0.f;
            const float inc = cb_rscan_add(H, lane);
            const float after = inc - H + suffix;
            const float ds = pre > 0.f ? dl * (e * T * gg - after) : 0.f;
            out[0] = w * gCs[0] * cr * (1.f - cr);
            out[1] = w * gCs[1] * cg * (1.f - cg);
            out[2] = w * gCs[2] * cb * (1.f - cb);
            out[3] = ds * (1.f - expf(-sg));
            suffix += __shfl(inc, 0);
        }
        if (ok) {
            float* o = a.d_head_grads + ((size_t)ray * N + i) * 9;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                o[k] = out[k];
                gmax = fmaxf(gmax, fabsf(out[k]));
            }
        }
    }
    if (a.d_gmax) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, d));
        // one atomic per wave, spread over NFL_GMAX_SLOTS words: every wave of the grid gets here at about the same
        // time, so a "look before you leap" load does not filter anything and same-address atomics serialise in L2
        // (non-negative floats order like their bit patterns; NaN/Inf gradients end up as a huge scale exponent
        // that nfl_loss_scale_from_bits clamps, and the NaNs themselves propagate to the result as in the reference)
        if (lane == 0 && gmax > 0.f)
            atomicMax(reinterpret_cast<unsigned*>(a.d_gmax) + (ray & (NFL_GMAX_SLOTS - 1)), __float_as_uint(gmax));
    }
}

__global__ __launch_bounds__(NFL_GMAX_SLOTS) void nfl_gmax_zero_kernel(float* g) { g[threadIdx.x] = 0.f; }

extern "C" int nfl_composite_backward(const nfl_compbwd_args* a, void* stream) {
    if (!a || !a->d_field_raw || !a->d_z || !a->d_head_grads) return NFL_EINVAL;
    if (a->n_rays < 0 || a->n_samples < 1 || a->n_samples > NFL_CB_MAXN) return NFL_EINVAL;
    // d_gmax is zeroed by a KERNEL, not hipMemsetAsync: inside a captured HIP graph the memset becomes a memset node, and with a
    // second process replaying graphs on the same GPU that node was seen to take effect out of order with the kernel that follows
    // it (the atomicMax results wiped: loss scale from an all-zero maximum, non-finite gradients)
    if (a->d_gmax) hipLaunchKernelGGL(nfl_gmax_zero_kernel, dim3(1), dim3(NFL_GMAX_SLOTS), 0, static_cast<hipStream_t>(stream), a->d_gmax);
    if (a->n_rays == 0) return NFL_OK;
    hipLaunchKernelGGL(nfl_compbwd_kernel, dim3((a->n_rays + 3) / 4), dim3(256), 0,
                       static_cast<hipStream_t>(stream), *a);
    return hipGetLastError() == hipSuccess ? NFL_OK : NFL_ELAUNCH;
}
