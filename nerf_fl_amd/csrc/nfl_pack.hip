// nfl_pack.hip -- repack fp32 nn.Linear parameters into the MFMA fragment stream.
//
// One wave (64 lanes) writes one k-step of one row tile: lane l = (h = l>>5,
// i = l&31) gathers the 8 weights W[row(i)][col(ks, h, j)], j = 0..7, converts
// them to fp16 (hi) and, in the 3-product mode, the fp16 residual (lo), and
// stores 16 B (+16 B) at its lane-linear position.  Replaces nothing in the
// reference (whose weights stay nn.Linear tensors, models/nerf.py:121-151); it is
// the layout step that lets the render kernel read W with ds_read_b128 and no
// swizzle.  HBM-bound and tiny: ~2.4 MB read, 1.2-2.4 MB written per field.
#include <hip/hip_runtime.h>

#include "../../include/nerf_fl_amd.h"
#include "nfl_plan.h"
#include "nfl_diag.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));

struct PackArgs {
    const NflPlan* plan;        // device copy
    nfl_field_params params;
    char* out;                  // packed buffer
    int32_t* status;            // NFL_STATUS_RANGE is OR-ed in when a weight exceeds fp16's range; may be null
};

// up to NFL_PACK_MAX_JOBS streams in one launch (a training step re-packs the forward and the dgrad stream of both
// fields after every optimizer update: four 8 us launches of pure latency became one)
struct PackBatch {
    int n;
    int first_block[NFL_PACK_MAX_JOBS + 1];
    PackArgs job[NFL_PACK_MAX_JOBS];
};

__global__ __launch_bounds__(64) void nfl_pack_kernel(const PackBatch B) {
    int jb = 0;
    while (jb + 1 < B.n && (int)blockIdx.x >= B.first_block[jb + 1]) ++jb;
    const PackArgs& a = B.job[jb];
    const NflPlan& P = *a.plan;
    const int lane = threadIdx.x;
    const int gks = blockIdx.x - B.first_block[jb];          // global k-step index in the stream
    if (gks >= P.total_ks) {
        // bias table: blocks total_ks .. total_ks + n_rt - 1, lanes 0..31
        const int t = gks - P.total_ks;
        if (t >= P.n_rt || lane >= 32) return;
        const NflRowTile& rt = P.rt[t];
        float v = 0.f;
        for (int b = 0; b < (rt.trans ? 0 : rt.nblk); ++b) {
            const int r = lane - rt.blk[b].dst_row;
            if (r >= 0 && r < rt.blk[b].nrows) v = a.params.bias[rt.blk[b].layer][rt.blk[b].src_row0 + r];
        }
        reinterpret_cast<float*>(a.out + P.bias_off)[t * 32 + lane] = v;
        return;
    }
    // locate the row tile containing this k-step: last t with rt[t].frag_off <= gks.  Binary search: every probe is
    // a dependent load from the plan in global memory, and a linear scan made this kernel 24 us of pure latency
    int t = 0;
    for (int hi = P.n_rt; hi - t > 1;) {
        const int mid = (t + hi) >> 1;
        if (P.rt[mid].frag_off <= gks) t = mid;
        else hi = mid;
    }
    const NflRowTile& rt = P.rt[t];
    int ks = gks - rt.frag_off;
    int s = 0;
    while (ks >= rt.seg[s].nks) { ks -= rt.seg[s].nks; ++s; }
    const NflSeg seg = rt.seg[s];

    const int i = lane & 31, h = lane >> 5;
    const float* wrow = nullptr;      // forward tiles: the source row of this lane
    const float* wcol = nullptr;      // dgrad tiles: the source column of this lane
    int ldt = 0;
    if (!rt.trans) {
        for (int b = 0; b < rt.nblk; ++b) {
            const int r = i - rt.blk[b].dst_row;
            if (r >= 0 && r < rt.blk[b].nrows) {
                const int L = rt.blk[b].layer;
                wrow = a.params.weight[L] + (size_t)(rt.blk[b].src_row0 + r) * P.ld[L];
            }
        }
    } else if (i < rt.tncols) {
        ldt = P.ld[seg.layer];
        wcol = a.params.weight[seg.layer] + rt.tcol0 + i;
    }
    float w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int m = seg.kind == NFL_SEG_ACT ? 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * h + (j & 3)
                                              : 16 * ks + 8 * h + j;
        w[j] = 0.f;
        if (m < seg.ncols) {
            if (wrow != nullptr) w[j] = wrow[seg.col0 + m];
            if (wcol != nullptr) w[j] = wcol[(size_t)(seg.col0 + m) * ldt];
        }
    }
    char* dst = a.out + (size_t)gks * P.ks_bytes + lane * 16;
    if (P.elem == 0 && a.status) {
        bool bad = false;
#pragma unroll
        for (int j = 0; j < 8; ++j) bad |= !(fabsf(w[j]) <= 65504.f);
        if (__builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) atomicOr(a.status, NFL_STATUS_RANGE);
    }
    if (P.elem == 0) {
        h8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            hi[j] = (_Float16)w[j];
            lo[j] = (_Float16)(w[j] - (float)hi[j]);
        }
#ifndef NFL_DIAG_RN_WT
        if (rt.trans && P.nsplit == 1) {
            // The single-image gradient chain (NFL_PREC_F16) multiplies by these fp16 transposed weights alone.  Rounded to
            // nearest, a weight's error persists for as long as the weight stays inside one fp16 interval, and the chain's
            // gradients carry it step after step -- the larger part of that mode's training-curve offset
            // (profiles/r03_psnr_backward_attribution.txt).  Rounded stochastically (v_cvt_sr_f16_f32, as in nfl_dgrad.hip) the
            // error is zero-mean, and because the draw is a hash of the weight's own fp32 bits and its place in the stream,
            // it is redrawn whenever the optimizer moves the weight: the errors of successive steps average out under Adam's
            // moments instead of adding up.  Deterministic in the weights; the forward streams are not touched.
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                unsigned r = __float_as_uint(w[j]) ^ (((unsigned)gks * 64u + (unsigned)lane) * 8u + (unsigned)j) * 0x9E3779B9u;
                r ^= r >> 16; r *= 0x7FEB352Du; r ^= r >> 15; r *= 0x846CA68Bu; r ^= r >> 16;
                unsigned o;
                asm("v_cvt_sr_f16_f32 %0, %1, %2" : "=v"(o) : "v"(w[j]), "v"(r));
                hi[j] = __builtin_bit_cast(_Float16, (unsigned short)o);
            }
        }
#endif
        *reinterpret_cast<h8*>(dst) = hi;
        if (P.nsplit == 3) *reinterpret_cast<h8*>(dst + 1024) = lo;
    } else {
        b8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            hi[j] = (__bf16)w[j];
            lo[j] = (__bf16)(w[j] - (float)hi[j]);
        }
        *reinterpret_cast<b8*>(dst) = hi;
        if (P.nsplit == 3) *reinterpret_cast<b8*>(dst + 1024) = lo;
    }
}

extern "C" int nfl_pack_fields(int32_t n_jobs, const nfl_pack_job* jobs, void* stream) {
    if (n_jobs < 0 || n_jobs > NFL_PACK_MAX_JOBS || (n_jobs > 0 && !jobs)) return NFL_EINVAL;
    if (n_jobs == 0) return NFL_OK;
    PackBatch B;
    B.n = n_jobs;
    int blocks = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const nfl_pack_job& J = jobs[j];
        const NflPlan* hp = static_cast<const NflPlan*>(J.h_plan);
        if (!hp || !J.d_plan || !J.params || !J.d_packed || hp->magic != NFL_PLAN_MAGIC) return NFL_EINVAL;
        if (J.packed_bytes < (size_t)hp->packed_bytes) return NFL_ESMALL;
        B.first_block[j] = blocks;
        blocks += hp->total_ks + hp->n_rt;
        B.job[j].plan = static_cast<const NflPlan*>(J.d_plan);
        B.job[j].params = *J.params;
        B.job[j].out = static_cast<char*>(J.d_packed);
        B.job[j].status = J.d_status;
    }
    B.first_block[n_jobs] = blocks;
    hipLaunchKernelGGL(nfl_pack_kernel, dim3(blocks), dim3(64), 0, static_cast<hipStream_t>(stream), B);
    return hipGetLastError() == hipSuccess ? NFL_OK : NFL_ELAUNCH;
}

extern "C" int nfl_pack_field(const void* h_plan, const void* d_plan, const nfl_field_params* params,
                              void* d_packed, size_t packed_bytes, int32_t* d_status, void* stream) {
    nfl_pack_job J;
    J.h_plan = h_plan;
    J.d_plan = d_plan;
    J.params = params;
    J.d_packed = d_packed;
    J.packed_bytes = packed_bytes;
    J.d_status = d_status;
    return nfl_pack_fields(1, &J, stream);
}
