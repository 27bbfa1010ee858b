// nfl_sample.hip -- hierarchical (inverse-CDF) importance sampling, one wave per ray.
//
// Replaces reference models/rendering.py:7-46 (sample_pdf) together with its call
// site :267-272 (mid-point bins, interior weights [:,1:-1], concat with the coarse
// depths and per-row sort).  Per ray: ~30 small ATen kernels become one wave doing
//   1. w = weights[1:-1] + 1e-5, wave reduction for the sum, pdf = w / sum
//   2. cdf = [0, inclusive wavefront prefix scan of pdf]      (shuffle scan, 64 bins per pass)
//      The arithmetic is the reference's CPU arithmetic, operation for operation, so that for the same coarse weights
//      the draws are bit-identical to torch's (tests/test_sample_pdf_gpu.py) -- which matters because sample_pdf is
//      discontinuous: a bin whose probability is within rounding of `eps` flips between two formulas (rendering.py:41-42).
//        - the sum follows ATen's fp32 CPU reduction order (SumKernel.cpp: 8-lane vectors, 4 interleaved vector
//          accumulators, the scalar tail, then the 8 partials in order), reproduced here by 32 lanes + one serial lane;
//        - the scan accumulates in fp64 and rounds to fp32 once per output, as ATen's CPU cumsum does; the pdf entries
//          lie in [2^-23, 1], so every fp64 partial sum is exact and the parallel scan order cannot show.
//   3. for every u: count of cdf <= u by binary search in LDS (searchsorted right=True),
//      clamp, gather, lerp; zero-width bins (< eps) get denominator 1
//   4. merge with the coarse depths by rank counting in LDS (stable; the result is the
//      sorted row the reference gets from torch.sort)
// Bound: HBM, algorithmic bytes per ray = 4*(2S + I (u) + S+I (out)); tiny next to the MLP.
#include <hip/hip_runtime.h>

#include "../../include/nerf_fl_amd.h"

#define NFL_SP_MAX 512   // max S + I per ray held in LDS

struct SampleArgs {
    const float* z; const float* w; const float* u; const float* u_row;
    int R, S, I;
    float* z_fine; float* samples;
};

__global__ __launch_bounds__(256) void nfl_sample_pdf_kernel(SampleArgs a) {
    __shared__ float lds[4][3 * NFL_SP_MAX];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ray = blockIdx.x * 4 + wave;
    if (ray >= a.R) return;
    const int S = a.S, I = a.I, M = S - 2, NB = S - 1, F = S + I;
    float* cdf = lds[wave];                 // [NB]
    float* bins = lds[wave] + NFL_SP_MAX;   // [NB]
    float* keys = lds[wave] + 2 * NFL_SP_MAX;   // [F]
    const float* zr = a.z + (size_t)ray * S;
    const float* wr = a.w + (size_t)ray * S + 1;
    const float eps = 1e-5f;

    // 1. sum of (w + eps) in ATen's CPU order.  Lane (k = lane / 8, v = lane % 8), k < 4, owns vector accumulator k,
    // element v: it adds the elements of vectors k, k + 4, k + 8, ... in order; left-over vectors go to accumulator 0.
    float total;
    if (M < 8) {           // ATen's scalar path for rows shorter than a vector: 4 interleaved scalar accumulators
        float acc = 0.f;
        const int nq = M >> 2;
        if (lane < 4)
            for (int i = 0; i < nq; ++i) acc += wr[(i << 2) + lane] + eps;
        if (lane == 0)
            for (int j = nq << 2; j < M; ++j) acc += wr[j] + eps;
        const float a1 = __shfl(acc, 1), a2 = __shfl(acc, 2), a3 = __shfl(acc, 3);
        total = __shfl(((acc + a1) + a2) + a3, 0);
    } else {
        const int nv = M >> 3, nfull = nv >> 2;          // whole 8-lane vectors; groups of 4 vectors
        const int k = lane >> 3, v = lane & 7;
        float acc = 0.f;
        if (lane < 32) {
            for (int i = 0; i < nfull; ++i) acc += wr[((i << 2) + k) * 8 + v] + eps;
            if (k == 0)
                for (int i = nfull << 2; i < nv; ++i) acc += wr[i * 8 + v] + eps;
        }
        // accumulator 0 += 1, += 2, += 3 (lanes 0..7)
        const float a1 = __shfl(acc, (lane & 7) + 8), a2 = __shfl(acc, (lane & 7) + 16), a3 = __shfl(acc, (lane & 7) + 24);
        acc = ((acc + a1) + a2) + a3;
        float fin = 0.f;
        for (int j = nv << 3; j < M; ++j) fin += wr[j] + eps;           // scalar tail
#pragma unroll
        for (int q = 0; q < 8; ++q) fin += __shfl(acc, q);              // the 8 partials, in order
        total = fin;
    }

    // 2. cdf by wavefront inclusive scan, 64 bins per pass (fp64 partial sums, see the header)
    double run = 0.0;
    if (lane == 0) cdf[0] = 0.f;
    for (int j0 = 0; j0 < M; j0 += 64) {
        const int j = j0 + lane;
        double v = j < M ? (double)((wr[j] + eps) / total) : 0.0;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const double n = __shfl_up(v, d);
            if (lane >= d) v += n;
        }
        v += run;
        if (j < M) cdf[j + 1] = (float)v;
        run = __shfl(v, 63);
    }
    for (int j = lane; j < NB; j += 64) bins[j] = 0.5f * (zr[j] + zr[j + 1]);
    for (int j = lane; j < S; j += 64) keys[j] = zr[j];
    // each wave owns its LDS region: a wave-level fence is all the ordering needed
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // 3. inverse CDF
    for (int i = lane; i < I; i += 64) {
        const float u = a.u ? a.u[(size_t)ray * I + i] : a.u_row[i];
        int lo = 0, hi = NB;                // first index with cdf > u  == count of cdf <= u
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
        }
        const int below = lo - 1 > 0 ? lo - 1 : 0;
        const int above = lo < M ? lo : M;
        const float c0 = cdf[below], c1 = cdf[above];
        const float b0 = bins[below], b1 = bins[above];
        float den = c1 - c0;
        if (den < eps) den = 1.f;
        const float smp = b0 + (u - c0) / den * (b1 - b0);
        keys[S + i] = smp;
        if (a.samples) a.samples[(size_t)ray * I + i] = smp;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // 4. sort by rank counting (all lanes read the same key: LDS broadcast)
    for (int e = lane; e < F; e += 64) {
        const float v = keys[e];
        int rank = 0;
        for (int k = 0; k < F; ++k) {
            const float o = keys[k];
            rank += (o < v || (o == v && k < e)) ? 1 : 0;
        }
        a.z_fine[(size_t)ray * F + rank] = v;
    }
}

extern "C" int nfl_sample_pdf(const float* d_z_coarse, const float* d_weights_coarse,
                              const float* d_u, const float* d_u_row,
                              int32_t n_rays, int32_t n_samples, int32_t n_importance,
                              float* d_z_fine, float* d_samples, void* stream) {
    if (!d_z_coarse || !d_weights_coarse || (!d_u && !d_u_row) || !d_z_fine) return NFL_EINVAL;
    if (n_rays < 0 || n_samples < 3 || n_importance < 1 || n_samples + n_importance > NFL_SP_MAX) return NFL_EINVAL;
    if (n_rays == 0) return NFL_OK;
    SampleArgs a{d_z_coarse, d_weights_coarse, d_u, d_u_row, n_rays, n_samples, n_importance, d_z_fine, d_samples};
    hipLaunchKernelGGL(nfl_sample_pdf_kernel, dim3((n_rays + 3) / 4), dim3(256), 0,
                       static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? NFL_OK : NFL_ELAUNCH;
}
