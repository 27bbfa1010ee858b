// nfl_wgrad.hip -- weight / bias gradients of the field MLP as streaming fp16 GEMMs.
//
//   dW_l[out, in] = sum over samples  delta_l[sample, out] * h_{l-1}[sample, in]
//   db_l[out]     = sum over samples  delta_l[sample, out]
// (what autograd's Linear backward does per point chunk in the reference, models/nerf.py
// 153-212).  Both operands come from the stashes the forward / dgrad kernels wrote: per
// 32-sample segment, per layer, MFMA-fragment-ordered fp16 with the SAMPLE on the lane (the
// gradients carry the pass' power-of-two loss scale, divided out at the flush).
// The contraction index here is the sample, so both operands have to be presented with
// the FEATURE on the lane and 8 consecutive samples in registers: the 1 KiB k-step images
// are DMA'd verbatim into LDS and read back with ds_read_b64_tr_b16 (hardware 4x16
// transpose), two reads per MFMA operand, no extra pass over the data.
//
// Composed through xyz_encoding_final.  That layer is linear (feat = W_fin h8 + b_fin, no activation), and its
// output is read by the direction layer and the first transient layer only, so
//     delta_feat = Wd^T delta_dirh (+ Wt^T delta_g1)        Wd = W_dir[:, :256], Wt = W_t0[:, :256]
// and with G = sum_s delta_dirh (x) h8  (128 x 256; Gt the same from delta_g1):
//     dW_fin        = Wd^T G (+ Wt^T Gt)            db_fin = Wd^T db_dir (+ Wt^T db_t0)
//     dW_dir[:, :256] = G W_fin^T + db_dir (x) b_fin    (dW_t0[:, :256] likewise from Gt)
// Neither `feat` nor `delta_feat` is ever stashed (16 KiB per 32-sample segment less written by the forward, 16 less
// by dgrad, 24 less read here); the stream accumulates G | Gt into a 256 x 256 scratch like any other job and
// nfl_wgrad_compose_kernel finishes the four products in fp32 (34-67 MFLOP, one launch).
//
// Roofline: HBM.  A 256x256 layer reads 32 KiB per segment for 2 x 64 MFMAs, 128 FLOP/B,
// far under the MFMA ridge, so the kernel is a stream (wg_body_rs below), with fp32 accumulators
// for the whole (<=256 x <=352) tile of dW in registers and one fp32 atomic flush per workgroup.
#include <hip/hip_runtime.h>
#include <string.h>

#include <type_traits>

#include "../../include/nerf_fl_amd.h"
#include "nfl_diag.h"
#include "nfl_plan.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

// segments in flight per wave, by pieces per wave (PW 4 / 5 / 6 / 8 with <= 6 in tiles / 8 with 8): what the register file holds
#ifndef WG_D4
#define WG_D4 8
#define WG_D5 8
#define WG_D6 7
#define WG_D8A 6
#define WG_D8B 5
#endif
#define WG_MAX_OT 8      // out tiles (32 features) per job
#define WG_MAX_IT 11     // in tiles per job
#define WG_NOT 2         // out tiles per wave
#define WG_SLOT (2048 * (WG_MAX_OT + WG_MAX_IT))

struct WgTile {
    int16_t slot;        // first k-step of the tile inside the segment record
    int16_t kind;        // NFL_SEG_ACT / NFL_SEG_NAT
    int16_t idx0;        // row0 / col0 of feature 0 of this tile in the weight
    int16_t nvalid;      // features of this tile that exist (<= 32)
};
#define WG_SCRATCH (-1)   // WgJob::layer of the job that accumulates G | Gt into WgArgs::scratch (256 x 256, ld 256)
struct WgJob {
    int32_t layer;       // NFL_P_* of the weight this job accumulates into, or WG_SCRATCH
    int32_t ld;          // row stride of that weight
    int32_t n_ot, n_it;
    int32_t n_wo, n_wi;  // waves across out tiles / in tiles (n_wo * n_wi == 4)
    int32_t do_bias;
    int32_t bias_layer_of_ot[WG_MAX_OT];   // each out tile may belong to another layer, weight and bias (-1: use `layer`)
    WgTile ot[WG_MAX_OT];
    WgTile it[WG_MAX_IT];
};
#define WG_MAX_JOBS 20
struct WgPlan {          // host-built once per (field, transient on/off); the caller keeps a device copy
    uint32_t magic;
    int32_t n_jobs;
    int32_t act_slots, grd_slots;
    int32_t w_numel[NFL_NUM_LAYERS], b_numel[NFL_NUM_LAYERS];   // sizes of the gradient tensors (0: layer absent)
    int32_t cost[WG_MAX_JOBS];     // 1 KiB pieces per wave per segment (4 / 5 / 6 / 8): the job's relative cost
    WgJob job[WG_MAX_JOBS];
};
struct WgArgs {
    const WgPlan* plan;  // device
    const char* act;     // activation stash
    const char* grd;     // gradient stash
    int n_seg;
    int act_rec, grd_rec;            // slots per segment record of the two stashes (twice the plan's with split stashes)
    int act_lo, grd_lo;              // split stashes: byte offset of the residual record behind the hi record (0: none)
    int slot_bytes;                  // bytes of one LDS slot of this launch (all hi (+ lo) pieces of the largest job)
    int wg_start[WG_MAX_JOBS + 1];   // workgroups [wg_start[j], wg_start[j+1]) work on job j
    nfl_field_grads g;
    float* scratch;      // (256, 256): rows 0..127 G (delta_dirh (x) h8), rows 128..255 Gt (delta_g1 (x) h8)
    // partial sums: workgroup `part` of job j stores its accumulators at partial + part_off[j] + part * part_len[j] (floats), as
    // [(wave * WG_NOT + a) * part_nitw[j] + b][r / 4][lane][r % 4] followed by the bias sums [(wave * WG_NOT + a)][lane]; nfl_wgrad_reduce_kernel
    // adds the parts up in a fixed order, divides by the loss scale and writes the gradient tensors
    float* partial;
    int part_off[WG_MAX_JOBS], part_len[WG_MAX_JOBS], part_nitw[WG_MAX_JOBS];
    int red_start[WG_MAX_JOBS + 1];  // reduction blocks [red_start[j], red_start[j+1]) belong to job j: one per (wave, a, b)
};
#define WG_MAX_WGS 256    // workgroups the partial-sum area is sized for (one per CU)

__device__ __forceinline__ int wg_orig(int kind, int i) {
    return kind == NFL_SEG_ACT ? 16 * (i >> 4) + 8 * ((i & 7) >> 2) + 4 * ((i >> 3) & 1) + (i & 3) : i;
}

#define WG_PSTRIDE 1056   // LDS stride of a 1 KiB k-step image: +32 B so the two k-steps of a tile fall on
                          // different banks for the transposed reads (4-way -> 2-way conflicts)
#define WG_TSTRIDE (2 * WG_PSTRIDE)
#undef WG_SLOT
#define WG_SLOT (WG_TSTRIDE * (WG_MAX_OT + WG_MAX_IT))

// feature-on-lane MFMA operand of MFMA k-step m (samples 16m..16m+15) from a tile image (2 k-step pieces)
__device__ __forceinline__ h8 wg_operand(const char* tile, int lane, int m) {
    // stash k-step image: [sample c][lane half h][8 values] = 32 B per sample.  A 16-lane group reads 4
    // samples x 16 features = 128 contiguous bytes; the two groups of a 32-lane half take the two k-steps of
    // the tile (pieces WG_PSTRIDE apart, 32 B off the 1 KiB grid): every read is bank-conflict free.
    const int g = lane >> 4, ip = lane & 15, q = ip >> 2, p = ip & 3;
    const int c = 16 * m + 8 * (g >> 1) + q;
    const char* ad = tile + (g & 1) * WG_PSTRIDE + c * 32 + p * 8;
    typedef __fp16 hf4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
    const h4 lo = __builtin_bit_cast(h4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) hf4*)(ad)));
    const h4 hi = __builtin_bit_cast(h4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) hf4*)(ad + 4 * 32)));
    h8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

template <int I0, int I1, class F>
__device__ __forceinline__ void wg_static_for(F&& f) {
    if constexpr (I0 < I1) {
        f(std::integral_constant<int, I0>{});
        wg_static_for<I0 + 1, I1>(f);
    }
}

template <int NITW>
__device__ __forceinline__ void wg_flush(const WgArgs& A, const int j, const int part, f16v (&acc)[WG_NOT][NITW], float (&bsum)[WG_NOT],
                                         const int wave, const int lane) {
    // flush: every workgroup stores its accumulators (still carrying the loss scale) as they are, 64 B per lane and tile; the
    // reduction kernel below knows which gradient element each of them is.  (Until round 3 this was 64 MB of fp32 atomics into
    // the gradient tensors at the end of the launch, 55 us during which the HBM idled, a zeroing and an unscaling launch around
    // it -- and a summation order that changed from run to run.)
    float* P = A.partial + (size_t)A.part_off[j] + (size_t)part * A.part_len[j];
    typedef float wg_f4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int a = 0; a < WG_NOT; ++a) {
#pragma unroll
        for (int b = 0; b < NITW; ++b) {
            // [tile][q][lane][4]: a store instruction (and a wave of the reduction) covers 1 KiB of contiguous memory
            float* dst = P + (size_t)((wave * WG_NOT + a) * NITW + b) * 1024 + lane * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<wg_f4*>(dst + q * 256) = wg_f4{acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]};
        }
        P[(size_t)4 * WG_NOT * NITW * 1024 + (wave * WG_NOT + a) * 64 + lane] = bsum[a];
    }
}

// One block per (job, wave, a, b) accumulator tile: sums the job's parts in order, unscales, and writes the elements the tile
// owns -- each gradient element is owned by exactly one tile of one job, so these are plain stores into tensors nobody zeroed.
__global__ __launch_bounds__(256) void nfl_wgrad_reduce_kernel(const WgArgs A, const float* gmax) {
    const WgPlan& P = *A.plan;
    int j = 0;
    while (j + 1 < P.n_jobs && (int)blockIdx.x >= A.red_start[j + 1]) ++j;
    const WgJob& J = P.job[j];
    const int nitw = A.part_nitw[j], t = blockIdx.x - A.red_start[j];
    const int b = t % nitw, wa = t / nitw, a = wa % WG_NOT, wave = wa / WG_NOT;
    const int wo = wave % J.n_wo, wi = wave / J.n_wo;
    const int lane = threadIdx.x & 63, rq = threadIdx.x >> 6;
    const int nparts = A.wg_start[j + 1] - A.wg_start[j];
    const int live = A.n_seg < nparts ? A.n_seg : nparts;        // workgroups beyond the segment count returned before their flush
    float inv;
    {
        unsigned v = 0u;
        for (int i = lane; i < NFL_GMAX_SLOTS; i += 64) {
            const unsigned o = reinterpret_cast<const unsigned*>(gmax)[i];
            v = o > v ? o : v;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const unsigned o = __shfl_xor(v, d);
            v = o > v ? o : v;
        }
        inv = 1.0f / nfl_loss_scale_from_bits(__builtin_amdgcn_readfirstlane(v));
    }
    const int ot = wo * WG_NOT + a;
    if (ot >= J.n_ot) return;
    const WgTile TO = J.ot[ot];
    const int layer = J.bias_layer_of_ot[ot] >= 0 ? J.bias_layer_of_ot[ot] : J.layer;
    const float* base = A.partial + (size_t)A.part_off[j];
    const int n = lane & 31, hh = lane >> 5;
    typedef float wg_f4 __attribute__((ext_vector_type(4)));
    float* W = layer == WG_SCRATCH ? A.scratch : A.g.weight[layer];
    if (wi + b * J.n_wi < J.n_it && W != nullptr) {
        const WgTile TI = J.it[wi + b * J.n_wi];
        const int on = wg_orig(TI.kind, n);
        wg_f4 sum = {0.f, 0.f, 0.f, 0.f};
        const float* src = base + (size_t)((wave * WG_NOT + a) * nitw + b) * 1024 + rq * 256 + lane * 4;
        // eight parts in flight at a time (a dependent chain of ~20 loads would cost the launch their latencies); the order is fixed
        const size_t pl = (size_t)A.part_len[j];
        int p = 0;
        for (; p + 8 <= live; p += 8) {
            wg_f4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const wg_f4*>(src + (size_t)(p + u) * pl);
            sum += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
        for (; p < live; ++p) sum += *reinterpret_cast<const wg_f4*>(src + (size_t)p * pl);
        if (on < TI.nvalid) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = 4 * rq + k;
                const int oi = wg_orig(TO.kind, (r & 3) + 8 * (r >> 2) + 4 * hh);
                if (oi < TO.nvalid) W[(size_t)(TO.idx0 + oi) * J.ld + TI.idx0 + on] = sum[k] * inv;
            }
        }
    }
    if (b == 0 && rq == 0 && J.do_bias && wi == 0 && layer != WG_SCRATCH && A.g.bias[layer] != nullptr) {
        const float* src = base + (size_t)4 * WG_NOT * nitw * 1024 + (wave * WG_NOT + a) * 64 + lane;
        float sb = 0.f;
        for (int p = 0; p < live; ++p) sb += src[(size_t)p * A.part_len[j]];
        const float tot = sb + __shfl_xor(sb, 32);
        const int oi = wg_orig(TO.kind, n);
        if (hh == 0 && oi < TO.nvalid) A.g.bias[layer][TO.idx0 + oi] = tot * inv;
    }
}

typedef unsigned wg_u4 __attribute__((ext_vector_type(4)));
// scalar base (the segment's record, uniform) + 32-bit per-lane offset: one VGPR of address per piece instead of two
__device__ __forceinline__ void wg_gload(wg_u4& dst, unsigned voff, const char* sbase) {
    asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(dst) : "v"(voff), "s"(sbase));
}
// wait until at most N vector-memory operations are outstanding; the register set rides through so its users stay below
template <int N, int PW>
__device__ __forceinline__ void wg_landed(wg_u4 (&r)[PW]) {
    static_assert(N < 64, "vmcnt is a 6-bit counter");
    static_assert(PW == 4 || PW == 5 || PW == 6 || PW == 8 || PW == 12 || PW == 16, "piece counts the dispatcher uses");
    if constexpr (PW == 4)
        asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "n"(N));
    else if constexpr (PW == 5)
        asm volatile("s_waitcnt vmcnt(%5)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]) : "n"(N));
    else if constexpr (PW == 6)
        asm volatile("s_waitcnt vmcnt(%6)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]) : "n"(N));
    else if constexpr (PW == 8)
        asm volatile("s_waitcnt vmcnt(%8)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]),
                     "+v"(r[6]), "+v"(r[7]) : "n"(N));
    else if constexpr (PW == 12)
        asm volatile("s_waitcnt vmcnt(%12)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]),
                     "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]) : "n"(N));
    else
        asm volatile("s_waitcnt vmcnt(%16)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]),
                     "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]),
                     "+v"(r[14]), "+v"(r[15]) : "n"(N));
}

// The stream is staged through REGISTERS.  An LDS-DMA ring (the first design: 4 slots of 32-40 KiB, 1.43 ms per
// fine pass) caps the bytes a workgroup keeps outstanding at what fits in the 160 KB of LDS, and this kernel waits
// for HBM ~60 % of the time, so bytes in flight are what counts.  Here every wave keeps D segments of its own
// 1 KiB pieces in flight in registers (nontemporal global_load_dwordx4, 4 VGPRs per piece; PW = pieces per wave
// per segment, NITW = in tiles per wave), copies the oldest set into a 2-slot LDS double buffer with
// ds_write_b128 when its turn comes, and re-issues loads into the freed registers at once: 1.13 ms = 5.7 TB/s.
// One barrier per segment; no branch inside a segment: a wave whose share is short works on a clamped
// (duplicate) tile and drops it at the flush.
// SPLIT (three-product weight gradient, NFL_PREC_F16X3): the residual images of every tile travel with the hi images (twice
// the pieces per segment, the lo pieces behind the hi pieces in the LDS slot) and every accumulator gets
// d_hi (x) h_hi + d_lo (x) h_hi + d_hi (x) h_lo -- one pass over hi + lo instead of three passes of the one-product GEMM.
template <int PW, int NITW, int D, bool SPLIT = false>
__device__ __forceinline__ void wg_body_rs(const WgArgs& A, const WgJob& J, const int act_slots, const int grd_slots,
                                           const int seg0, const int seg1, const int seg_first, const int seg_step, const int jidx, char* smem) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wo = wave % J.n_wo, wi = wave / J.n_wo;
    const int n_hi = 2 * (J.n_ot + J.n_it);
    const int n_pieces = SPLIT ? 2 * n_hi : n_hi;

    unsigned poff[PW];          // byte offset of this lane's 16 B of the piece inside its segment record
    unsigned outmask = 0u;      // bit pp: piece pp comes from the gradient stash (else: activation stash)
    int pdst[PW];
#pragma unroll
    for (int pp = 0; pp < PW; ++pp) {
        int p = wave + 4 * pp;           // (wave * PW + pp, a wave's pieces contiguous in the record, runs at the same speed: measured)
        p = p < n_pieces ? p : n_pieces - 1;
        const bool lo = p >= n_hi;
        const int q = lo ? p - n_hi : p;
        const int t = q >> 1;
        const bool is_out = t < J.n_ot;
        const int slot = (is_out ? J.ot[t].slot : J.it[t - J.n_ot].slot) + (q & 1);
        poff[pp] = (unsigned)slot * 1024u + lane * 16u + (lo ? (unsigned)(is_out ? A.grd_lo : A.act_lo) : 0u);
        outmask |= is_out ? (1u << pp) : 0u;
        pdst[pp] = p * WG_PSTRIDE + lane * 16;
    }
    outmask = __builtin_amdgcn_readfirstlane(outmask);
    const size_t grd_stride = (size_t)grd_slots * 1024, act_stride = (size_t)act_slots * 1024;
    // The loads and their waits are hand-issued: left to hipcc the loop header gets an s_waitcnt vmcnt(0), which
    // drains all D segments once per trip.  VMEM operations return in order, so "all but the (D-1)*PW youngest"
    // is exactly "the oldest register set has landed"; nothing else in the loop touches vector memory.
    wg_u4 R[D][PW];
    auto gload = [&](int seg, auto DD) __attribute__((always_inline)) {
        constexpr int d = decltype(DD)::value;
        // logical segment `seg` of this workgroup is record seg_first + seg * seg_step of the stashes (see the kernel below)
        const int sg = seg_first + (seg < seg1 ? seg : seg1 - 1) * seg_step;          // surplus loads re-read the last segment
        const char* bo = A.grd + (size_t)sg * grd_stride;
        const char* bi = A.act + (size_t)sg * act_stride;
#pragma unroll
        for (int pp = 0; pp < PW; ++pp) wg_gload(R[d][pp], poff[pp], ((outmask >> pp) & 1u) ? bo : bi);
    };

    int my_ot[WG_NOT], my_it[NITW];
#pragma unroll
    for (int a = 0; a < WG_NOT; ++a) my_ot[a] = wo * WG_NOT + a < J.n_ot ? wo * WG_NOT + a : J.n_ot - 1;
#pragma unroll
    for (int b = 0; b < NITW; ++b) my_it[b] = wi + b * J.n_wi < J.n_it ? wi + b * J.n_wi : J.n_it - 1;

    f16v acc[WG_NOT][NITW];
#pragma unroll
    for (int a = 0; a < WG_NOT; ++a)
#pragma unroll
        for (int b = 0; b < NITW; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float bsum[WG_NOT] = {0.f, 0.f};
    // tile offsets inside a slot, in registers: J lives in global memory and the asm memory clobbers below would
    // otherwise make the loop re-load J.n_ot (a vmcnt(0) wait per segment)
    int ot_off[WG_NOT], it_off[NITW];
#pragma unroll
    for (int a = 0; a < WG_NOT; ++a) ot_off[a] = my_ot[a] * WG_TSTRIDE;
#pragma unroll
    for (int b = 0; b < NITW; ++b) it_off[b] = (J.n_ot + my_it[b]) * WG_TSTRIDE;

#ifdef NFL_DIAG_WGRAD_LDSDMA
    // timing ablation (nfl_diag.h): the same pieces of the same segments, fetched by LDS-DMA (global_load_lds, 16 B per lane) straight
    // into the two LDS slots instead of through registers; nothing consumes them, nothing is flushed -- what does THAT stream sustain?
    if (A.n_seg >= 0) {
        auto dma = [&](int seg) __attribute__((always_inline)) {
            const int sg = seg_first + (seg < seg1 ? seg : seg1 - 1) * seg_step;
            const char* bo = A.grd + (size_t)sg * grd_stride;
            const char* bi = A.act + (size_t)sg * act_stride;
            char* slot = smem + ((seg - seg0) & 1) * A.slot_bytes;
#pragma unroll
            for (int pp = 0; pp < PW; ++pp) {
                const char* src = (((outmask >> pp) & 1u) ? bo : bi) + poff[pp];
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(slot + (pdst[pp] - lane * 16)), 16, 0, 0);
            }
        };
        dma(seg0);
        dma(seg0 + 1);
        for (int seg = seg0; seg < seg1; ++seg) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");      // all but the youngest segment's pieces: `seg` has landed
            dma(seg + 2);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
#endif
    wg_static_for<0, D>([&](auto DD) __attribute__((always_inline)) { gload(seg0 + decltype(DD)::value, DD); });
    for (int base = seg0; base < seg1; base += D) {
        wg_static_for<0, D>([&](auto DD) __attribute__((always_inline)) {
            constexpr int d = decltype(DD)::value;
            const int seg = base + d;
            if (seg < seg1) {                                     // uniform
                char* slot = smem + ((seg - seg0) & 1) * A.slot_bytes;
                wg_landed<(D - 1) * PW, PW>(R[d]);
#ifdef NFL_DIAG_WGRAD_LOADS_ONLY
                // timing ablation (nfl_diag.h): the load stream alone -- no LDS round trip, no barrier, no MFMA; results wrong by construction
                if (A.n_seg >= 0) {
#pragma unroll
                    for (int pp = 0; pp < PW; ++pp) bsum[0] += __uint_as_float(R[d][pp].x ^ R[d][pp].y ^ R[d][pp].z ^ R[d][pp].w);
                    gload(seg + D, DD);
                    return;
                }
#endif
#pragma unroll
                for (int pp = 0; pp < PW; ++pp) *reinterpret_cast<wg_u4*>(slot + pdst[pp]) = R[d][pp];
                gload(seg + D, DD);
                __builtin_amdgcn_sched_barrier(0);
                // raw barrier: __syncthreads() carries a fence that hipcc lowers to s_waitcnt vmcnt(0), which would
                // drain the D segments in flight; only the LDS writes have to be complete here
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                const char* base_l = slot;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    h8 av[WG_NOT], bv[NITW];
#pragma unroll
                    for (int a = 0; a < WG_NOT; ++a) av[a] = wg_operand(base_l + ot_off[a], lane, m);
#pragma unroll
                    for (int b = 0; b < NITW; ++b) bv[b] = wg_operand(base_l + it_off[b], lane, m);
#pragma unroll
                    for (int a = 0; a < WG_NOT; ++a) {
                        float sacc = 0.f;
#pragma unroll
                        for (int j = 0; j < 8; ++j) sacc += (float)av[a][j];
                        bsum[a] += sacc;
                    }
#pragma unroll
                    for (int b = 0; b < NITW; ++b)
#pragma unroll
                        for (int a = 0; a < WG_NOT; ++a)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[a], bv[b], acc[a][b], 0, 0, 0);
                    if constexpr (SPLIT) {
                        const char* base_lo = base_l + n_hi * WG_PSTRIDE;        // the lo pieces follow the hi pieces
                        h8 al[WG_NOT];
#pragma unroll
                        for (int a = 0; a < WG_NOT; ++a) {
                            al[a] = wg_operand(base_lo + ot_off[a], lane, m);
                            float sacc = 0.f;
#pragma unroll
                            for (int j = 0; j < 8; ++j) sacc += (float)al[a][j];
                            bsum[a] += sacc;                              // db = sum_s (d_hi + d_lo)
                        }
#pragma unroll
                        for (int b = 0; b < NITW; ++b)
#pragma unroll
                            for (int a = 0; a < WG_NOT; ++a)
                                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[a], bv[b], acc[a][b], 0, 0, 0);
#pragma unroll
                        for (int b = 0; b < NITW; ++b) {
                            const h8 bl = wg_operand(base_lo + it_off[b], lane, m);
#pragma unroll
                            for (int a = 0; a < WG_NOT; ++a)
                                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[a], bl, acc[a][b], 0, 0, 0);
                        }
                    }
                }
            }
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the surplus loads still target live registers
#ifdef NFL_DIAG_WGRAD_NOFLUSH
    if (A.n_seg >= 0) {       // timing ablation (nfl_diag.h): keep the accumulators alive, skip the atomics -- results wrong by construction
        float keep = 0.f;
#pragma unroll
        for (int a = 0; a < WG_NOT; ++a)
#pragma unroll
            for (int b = 0; b < NITW; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) keep += acc[a][b][r];
        if (keep == 123.456f) A.scratch[0] = keep;
        return;
    }
#endif
    wg_flush<NITW>(A, jidx, (int)blockIdx.x - A.wg_start[jidx], acc, bsum, wave, lane);
}

template <bool SPLIT>
__global__ __launch_bounds__(256, 1) void nfl_wgrad_kernel(const WgArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const WgPlan& P = *A.plan;
    int j = 0;
    while (j + 1 < P.n_jobs && (int)blockIdx.x >= A.wg_start[j + 1]) ++j;
    const int part = blockIdx.x - A.wg_start[j], nparts = A.wg_start[j + 1] - A.wg_start[j];
    // The job's workgroups take its segments ROUND-ROBIN (workgroup `part` of `nparts`: records part, part + nparts, ...), not one
    // contiguous range each: every job crosses the stashes at the same pace (that is what the dealing equalises), so all 256
    // workgroups then read the same neighbourhood of memory at any time.  Same-box A/B against contiguous ranges
    // (-DWG_SEGMENTS_CONTIGUOUS): 0.918 vs 0.934 ms per fine pass, the split kernel of f16x3 2.10 vs 2.21
#ifdef WG_SEGMENTS_CONTIGUOUS
    const int seg_first = (int)((long long)A.n_seg * part / nparts), seg_step = 1;
    const int seg0 = 0, seg1 = (int)((long long)A.n_seg * (part + 1) / nparts) - seg_first;
#else
    const int seg_first = part, seg_step = nparts;
    const int seg0 = 0, seg1 = part < A.n_seg ? (A.n_seg - part + nparts - 1) / nparts : 0;
#endif
    if (seg0 >= seg1) return;
    const WgJob& J = P.job[j];
    const int pw = P.cost[j];
    const int nitw = (J.n_it + J.n_wi - 1) / J.n_wi;       // in tiles per wave
    if constexpr (SPLIT) {      // twice the pieces per segment in flight per set: fewer sets (the registers are the same)
        if (pw <= 4) {
            if (nitw <= 1) wg_body_rs<8, 1, 4, true>(A, J, A.act_rec, A.grd_rec, seg0, seg1, seg_first, seg_step, j, smem);
            else wg_body_rs<8, 2, 4, true>(A, J, A.act_rec, A.grd_rec, seg0, seg1, seg_first, seg_step, j, smem);
        } else if (pw <= 5) {       // 10 pieces per wave would do; that instantiation gave wrong sums on the GPU (not understood), 12 is verified
            wg_body_rs<12, 2, 3, true>(A, J, A.act_rec, A.grd_rec, seg0, seg1, seg_first, seg_step, j, smem);
        } else if (pw <= 6) {
            wg_body_rs<12, 4, 3, true>(A, J, A.act_rec, A.grd_rec, seg0, seg1, seg_first, seg_step, j, smem);
        } else {
            if (nitw <= 5) wg_body_rs<16, 5, 2, true>(A, J, A.act_rec, A.grd_rec, seg0, seg1, seg_first, seg_step, j, smem);
            else if (nitw <= 6) wg_body_rs<16, 6, 2, true>(A, J, A.act_rec, A.grd_rec, seg0, seg1, seg_first, seg_step, j, smem);
            else wg_body_rs<16, 8, 2, true>(A, J, A.act_rec, A.grd_rec, seg0, seg1, seg_first, seg_step, j, smem);
        }
        return;
    }
    if (pw <= 4) {
        if (nitw <= 1) wg_body_rs<4, 1, WG_D4>(A, J, A.act_rec, A.grd_rec, seg0, seg1, seg_first, seg_step, j, smem);
        else wg_body_rs<4, 2, WG_D4>(A, J, A.act_rec, A.grd_rec, seg0, seg1, seg_first, seg_step, j, smem);
    } else if (pw <= 5) {
        wg_body_rs<5, 2, WG_D5>(A, J, A.act_rec, A.grd_rec, seg0, seg1, seg_first, seg_step, j, smem);
    } else if (pw <= 6) {          // G of a pass without the transient head: 4 out x 8 in tiles
        wg_body_rs<6, 4, WG_D6>(A, J, A.act_rec, A.grd_rec, seg0, seg1, seg_first, seg_step, j, smem);
    } else {
        if (nitw <= 5) wg_body_rs<8, 5, WG_D8A>(A, J, A.act_rec, A.grd_rec, seg0, seg1, seg_first, seg_step, j, smem);
        else if (nitw <= 6) wg_body_rs<8, 6, WG_D8A>(A, J, A.act_rec, A.grd_rec, seg0, seg1, seg_first, seg_step, j, smem);
        else wg_body_rs<8, 8, WG_D8B>(A, J, A.act_rec, A.grd_rec, seg0, seg1, seg_first, seg_step, j, smem);
    }
}

// Whole-tensor passes around the GEMM kernel: op 0 zeroes the gradient tensors, op 1 divides them by the loss
// scale.  blockIdx.y = tensor (weights then biases), blockIdx.x strides over its elements.
#define WG_NTENS (2 * NFL_NUM_LAYERS + 1)     // weights, biases, the composition scratch
struct WgTensors {
    float* ptr[WG_NTENS];
    int n[WG_NTENS];
};

// ---------------------------------------------------------------------------------
// The four small fp32 products of the composition (file header), one launch: task t computes
//   C[m, n] = sum_k A[m, k] B[k, n] (+ sum_k A2[m, k] B2[k, n]) (+ u[m] v[n])
// over element strides of which one per operand is 1, 32 x 32 output tiles per 256-thread workgroup.
// The operands are the fp32 master weights and the finished (unscaled) G / bias gradients; everything is L2-resident.
struct WgGemm {
    const float* A; int sam, sak;
    const float* B; int sbk, sbn;
    const float* A2; int sam2, sak2;
    const float* B2; int sbk2, sbn2;
    const float* u; const float* v;       // rank-1 term (NULL: none)
    const float* addm;                    // + addm[m] (NULL: none)
    int avec, bvec;                       // the k-contiguous operand may be fetched with 16-byte loads (rows 16-byte aligned)
    float* Cp; int ldc;
    int M, N, K, K2;                      // K2 = 0: no second product
    int tiles_n, tile0;                   // tiles across N; index of this task's first workgroup
};
#define WG_MAX_GEMM 4
struct WgCompose {
    int n_tasks, n_wg;
    WgGemm t[WG_MAX_GEMM];
};
// One 32 x 32 output tile per workgroup on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact products, fp32
// accumulation): the four waves split K, every lane fetches its operands straight from global memory (float4 along k where
// k is the contiguous index, one dword per k otherwise; all loads of a wave are in flight before its first MFMA), the
// partial tiles meet in LDS.  MFMA j of a wave takes, in lane half h, k = k_wave + 8 (j / 4) + 4 h + (j % 4): any pairing
// of the two k of a step is as good as another as long as A and B agree.  (The first version staged 32 x 128 operand
// blocks through LDS for scalar FMAs and was LDS-bound: 20-30 us; this one is one memory round trip and 16-32 MFMAs per wave.)
typedef float wg_f4 __attribute__((ext_vector_type(4)));
template <int NJ>       // MFMAs per wave = (K / 4) / 2
__device__ __forceinline__ void wg_compose_pass(f16v& acc, const float* A, int sam, int sak, const float* B, int sbk, int sbn,
                                                int m, int n, bool n_ok, int k_wave, int h, bool avec, bool bvec) {
    float ra[NJ], rb[NJ];
#pragma unroll
    for (int g = 0; g < NJ / 4; ++g) {
        const int kq = k_wave + 8 * g + 4 * h;
        if (sak == 1 && avec) {
            const wg_f4 v = *reinterpret_cast<const wg_f4*>(A + (size_t)m * sam + kq);
            ra[4 * g] = v[0]; ra[4 * g + 1] = v[1]; ra[4 * g + 2] = v[2]; ra[4 * g + 3] = v[3];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) ra[4 * g + e] = A[(size_t)m * sam + (size_t)(kq + e) * sak];
        }
        if (!n_ok) {
#pragma unroll
            for (int e = 0; e < 4; ++e) rb[4 * g + e] = 0.f;
        } else if (sbk == 1 && sbn != 1 && bvec) {
            const wg_f4 v = *reinterpret_cast<const wg_f4*>(B + (size_t)n * sbn + kq);
            rb[4 * g] = v[0]; rb[4 * g + 1] = v[1]; rb[4 * g + 2] = v[2]; rb[4 * g + 3] = v[3];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) rb[4 * g + e] = B[(size_t)(kq + e) * sbk + (size_t)n * sbn];
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[j], rb[j], acc, 0, 0, 0);
}

__global__ __launch_bounds__(256) void nfl_wgrad_compose_kernel(const WgCompose Cc) {
    __shared__ float red[4][16][64];
    int ti = 0;
    while (ti + 1 < Cc.n_tasks && (int)blockIdx.x >= Cc.t[ti + 1].tile0) ++ti;
    const WgGemm& T = Cc.t[ti];
    const int tile = blockIdx.x - T.tile0, m0 = 32 * (tile / T.tiles_n), n0 = 32 * (tile % T.tiles_n);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5;
    const int m = m0 + (lane & 31), n = n0 + (lane & 31);      // M is a multiple of 32 (256 or 128); N is 256 or 1
    const bool n_ok = n < T.N;
    f16v acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // K and K2 are 128 or 256 (host-checked): 32 or 64 per wave
    if (T.K == 256) wg_compose_pass<32>(acc, T.A, T.sam, T.sak, T.B, T.sbk, T.sbn, m, n, n_ok, 64 * wave, h, T.avec, T.bvec);
    else wg_compose_pass<16>(acc, T.A, T.sam, T.sak, T.B, T.sbk, T.sbn, m, n, n_ok, 32 * wave, h, T.avec, T.bvec);
    if (T.K2 == 256) wg_compose_pass<32>(acc, T.A2, T.sam2, T.sak2, T.B2, T.sbk2, T.sbn2, m, n, n_ok, 64 * wave, h, T.avec, T.bvec);
    else if (T.K2 == 128) wg_compose_pass<16>(acc, T.A2, T.sam2, T.sak2, T.B2, T.sbk2, T.sbn2, m, n, n_ok, 32 * wave, h, T.avec, T.bvec);
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = 4 * wave + q;                             // accumulator register r: row 8 (r / 4) + 4 h + (r % 4), column lane % 32
        const float c = red[0][r][lane] + red[1][r][lane] + red[2][r][lane] + red[3][r][lane];
        const int row = m0 + 8 * (r >> 2) + 4 * h + (r & 3);
        if (n_ok) T.Cp[(size_t)row * T.ldc + n] = (T.u ? __builtin_fmaf(T.u[row], T.v[n], c) : c) + (T.addm ? T.addm[row] : 0.f);
    }
}
__global__ __launch_bounds__(256) void nfl_wgrad_scale_kernel(const WgTensors T, const int op, const float* gmax) {
    float* p = T.ptr[blockIdx.y];
    const int n = T.n[blockIdx.y];
    if (p == nullptr || n == 0) return;
    float f = 0.f;
    if (op != 0) {
        unsigned v = 0u;
        for (int i = threadIdx.x & 63; i < NFL_GMAX_SLOTS; i += 64) {
            const unsigned o = reinterpret_cast<const unsigned*>(gmax)[i];
            v = o > v ? o : v;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const unsigned o = __shfl_xor(v, d);
            v = o > v ? o : v;
        }
        f = 1.0f / nfl_loss_scale_from_bits(__builtin_amdgcn_readfirstlane(v));
    }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) p[i] = op == 0 ? 0.f : p[i] * f;
}

// ---------------------------------------------------------------------------------
static void add_tiles(WgTile* dst, int& n, int slot0, int kind, int idx0, int count) {
    for (int t = 0; 32 * t < count; ++t) {
        WgTile w;
        w.slot = (int16_t)(slot0 + 2 * t);
        w.kind = (int16_t)kind;
        w.idx0 = (int16_t)(idx0 + 32 * t);
        w.nvalid = (int16_t)(count - 32 * t < 32 ? count - 32 * t : 32);
        dst[n++] = w;
    }
}
static WgJob make_job(int layer, int ld, bool bias) {
    WgJob j;
    memset(&j, 0, sizeof(j));
    j.layer = layer;
    j.ld = ld;
    j.do_bias = bias ? 1 : 0;
    for (int i = 0; i < WG_MAX_OT; ++i) j.bias_layer_of_ot[i] = -1;
    return j;
}
static void finish_job(WgJob& j) {
    if (j.n_ot > 4) { j.n_wo = 4; j.n_wi = 1; }          // 2 out tiles per wave, all in tiles
    else if (j.n_ot > 2) { j.n_wo = 2; j.n_wi = 2; }
    else { j.n_wo = 1; j.n_wi = 4; }                     // heads: split the in tiles
}

extern "C" size_t nfl_wgrad_plan_bytes(void) { return sizeof(WgPlan); }

extern "C" int nfl_wgrad_plan_build(const nfl_field_desc* d, int32_t use_transient, void* h_plan, size_t bytes) {
    NflPlan p;
    if (!d || !h_plan) return NFL_EINVAL;
    if (bytes < sizeof(WgPlan)) return NFL_ESMALL;
    if (nfl_plan_fill(d, NFL_PREC_F16X3, &p) != NFL_OK) return NFL_EINVAL;
    const int nkp = p.nkp, cx = 6 * d->n_emb_xyz + 3, cd = 6 * d->n_emb_dir + 3, W = NFL_W, H = NFL_W / 2;
    const bool ut = p.has_t && use_transient;
    WgPlan& P = *static_cast<WgPlan*>(h_plan);
    memset(&P, 0, sizeof(P));
    P.magic = NFL_PLAN_MAGIC ^ 0x57u;
    P.act_slots = nfl_act_slots(nkp);
    P.grd_slots = NFL_GRD_SLOTS;
    int nj = 0;
    auto push = [&](WgJob j) {
        finish_job(j);
        const int pw = (2 * (j.n_ot + j.n_it) + 3) / 4;
        const int nitw = (j.n_it + j.n_wi - 1) / j.n_wi;
        P.cost[nj] = pw <= 4 ? 4 : (pw <= 5 ? 5 : ((pw <= 6 && nitw <= 4) ? 6 : 8));
        P.job[nj++] = j;
    };
    for (int l = 1; l <= 8; ++l) {
        WgJob j = make_job(NFL_P_XYZ1 + l - 1, p.ld[NFL_P_XYZ1 + l - 1], true);
        add_tiles(j.ot, j.n_ot, NFL_GRD_D(l), NFL_SEG_ACT, 0, W);
        if (l == 1 || l == 5) {
            add_tiles(j.it, j.n_it, 0, NFL_SEG_NAT, 0, cx);
            if (l == 5) {
                push(j);
                j = make_job(NFL_P_XYZ1 + 4, p.ld[NFL_P_XYZ1 + 4], false);
                add_tiles(j.ot, j.n_ot, NFL_GRD_D(5), NFL_SEG_ACT, 0, W);
                add_tiles(j.it, j.n_it, nfl_act_h(nkp, 4), NFL_SEG_ACT, cx, W);
            }
        } else {
            add_tiles(j.it, j.n_it, nfl_act_h(nkp, l - 1), NFL_SEG_ACT, 0, W);
        }
        push(j);
    }
    if (ut) {   // sigma head (without the transient head it shares the h8 stream of the next job)
        WgJob j = make_job(NFL_P_SIGMA, W, true);
        add_tiles(j.ot, j.n_ot, NFL_GRD_HEADS + 0, NFL_SEG_NAT, 0, 1);
        add_tiles(j.it, j.n_it, nfl_act_h(nkp, 8), NFL_SEG_ACT, 0, W);
        push(j);
    }
    {   // G | Gt = (delta_dirh | delta_g1) (x) h8 into the scratch: everything that touches `feat` (xyz_encoding_final
        // itself and the first 256 input columns of dir_encoding / transient_encoding.0) is composed from it.
        // Without the transient head there is room for a fifth out tile: the sigma head reads the same h8 (ld 256 too)
        WgJob j = make_job(WG_SCRATCH, W, !ut);
        add_tiles(j.ot, j.n_ot, NFL_GRD_DIRH, NFL_SEG_ACT, 0, H);
        if (ut) {
            add_tiles(j.ot, j.n_ot, NFL_GRD_G(1), NFL_SEG_ACT, H, H);
        } else {
            j.bias_layer_of_ot[j.n_ot] = NFL_P_SIGMA;
            add_tiles(j.ot, j.n_ot, NFL_GRD_HEADS + 0, NFL_SEG_NAT, 0, 1);
        }
        add_tiles(j.it, j.n_it, nfl_act_h(nkp, 8), NFL_SEG_ACT, 0, W);
        push(j);
    }
    {   // dir_encoding: the side inputs [dir PE | appearance] (columns 256..) and the bias
        WgJob j = make_job(NFL_P_DIR, p.ld[NFL_P_DIR], true);
        add_tiles(j.ot, j.n_ot, NFL_GRD_DIRH, NFL_SEG_ACT, 0, H);
        add_tiles(j.it, j.n_it, nfl_act_d(nkp), NFL_SEG_NAT, W, cd);
        if (p.has_a) add_tiles(j.it, j.n_it, nfl_act_d(nkp) + 2, NFL_SEG_NAT, W + cd, p.n_a);
        push(j);
    }
    {   // rgb head
        WgJob j = make_job(NFL_P_RGB, H, true);
        add_tiles(j.ot, j.n_ot, NFL_GRD_HEADS + 1, NFL_SEG_NAT, 0, 3);
        add_tiles(j.it, j.n_it, nfl_act_dirh(nkp), NFL_SEG_ACT, 0, H);
        push(j);
    }
    if (ut) {
        {
            WgJob j = make_job(NFL_P_T0, p.ld[NFL_P_T0], true);
            add_tiles(j.ot, j.n_ot, NFL_GRD_G(1), NFL_SEG_ACT, 0, H);
            add_tiles(j.it, j.n_it, nfl_act_tau(nkp), NFL_SEG_NAT, W, d->n_tau);      // the transient code (columns 256..) and the bias
            push(j);
        }
        for (int m = 2; m <= 4; ++m) {
            WgJob j = make_job(NFL_P_T0 + m - 1, H, true);
            add_tiles(j.ot, j.n_ot, NFL_GRD_G(m), NFL_SEG_ACT, 0, H);
            add_tiles(j.it, j.n_it, nfl_act_g(nkp, m - 1), NFL_SEG_ACT, 0, H);
            push(j);
        }
        {   // the three transient heads share the g4 stream: one out "tile" each
            WgJob j = make_job(NFL_P_TSIGMA, H, true);
            add_tiles(j.ot, j.n_ot, NFL_GRD_HEADS + 2, NFL_SEG_NAT, 0, 1);
            add_tiles(j.ot, j.n_ot, NFL_GRD_HEADS + 3, NFL_SEG_NAT, 0, 3);
            add_tiles(j.ot, j.n_ot, NFL_GRD_HEADS + 4, NFL_SEG_NAT, 0, 1);
            j.bias_layer_of_ot[0] = NFL_P_TSIGMA;
            j.bias_layer_of_ot[1] = NFL_P_TRGB;
            j.bias_layer_of_ot[2] = NFL_P_TBETA;
            add_tiles(j.it, j.n_it, nfl_act_g(nkp, 4), NFL_SEG_ACT, 0, H);
            push(j);
        }
    }
    P.n_jobs = nj;
    for (int L = 0; L < NFL_NUM_LAYERS; ++L) {
        const bool tr = L >= NFL_P_T0;
        const int rows = (L <= NFL_P_FINAL) ? W : (L == NFL_P_DIR || (L >= NFL_P_T0 && L < NFL_P_T0 + 4)) ? H
                         : (L == NFL_P_RGB || L == NFL_P_TRGB) ? 3 : 1;
        const bool present = tr ? p.has_t != 0 : true;      // transient layers of the model that this pass does not use get zero gradients
        P.w_numel[L] = present ? rows * p.ld[L] : 0;
        P.b_numel[L] = present ? rows : 0;
    }
    return NFL_OK;
}

extern "C" size_t nfl_wgrad_scratch_bytes(void) {
    // G (256 x 256) + the partial sums of at most WG_MAX_WGS workgroups with the largest accumulator set (2 x 8 tiles per wave)
    return ((size_t)NFL_W * NFL_W + (size_t)WG_MAX_WGS * (4 * WG_NOT * 8 * 1024 + 4 * WG_NOT * 64)) * sizeof(float);
}

extern "C" int nfl_mlp_wgrad(const void* h_wplan, const void* d_wplan, const char* d_act_stash,
                             const char* d_grad_stash, const float* d_gmax, int32_t n_rays, int32_t n_samples,
                             int32_t bwd_prec, const nfl_field_params* params, float* d_scratch,
                             const nfl_field_grads* grads, void* stream) {
    const WgPlan* hp = static_cast<const WgPlan*>(h_wplan);
    if (!hp || hp->magic != (NFL_PLAN_MAGIC ^ 0x57u) || !d_wplan || !d_act_stash || !d_grad_stash || !d_gmax || !grads
        || !params || !d_scratch)
        return NFL_EINVAL;
    // the composition reads these master weights; the side-input job of a layer needs its bias gradient as well
    if (!params->weight[NFL_P_FINAL] || !params->bias[NFL_P_FINAL] || !params->weight[NFL_P_DIR]) return NFL_EINVAL;
    const bool ut = hp->n_jobs > 0 && hp->job[hp->n_jobs - 1].layer == NFL_P_TSIGMA;
    if (ut && !params->weight[NFL_P_T0]) return NFL_EINVAL;
    if ((grads->weight[NFL_P_DIR] || grads->weight[NFL_P_FINAL]) && !grads->bias[NFL_P_DIR]) return NFL_EINVAL;
    if (ut && (grads->weight[NFL_P_T0] || grads->weight[NFL_P_FINAL]) && !grads->bias[NFL_P_T0]) return NFL_EINVAL;
    if (n_rays < 0 || n_samples < 1) return NFL_EINVAL;
    if (bwd_prec != NFL_PREC_F16 && bwd_prec != NFL_PREC_F16W && bwd_prec != NFL_PREC_F16X3) return NFL_EINVAL;
    const int mult = bwd_prec == NFL_PREC_F16X3 ? 2 : 1;       // split stashes: [hi record | lo record] per segment
    WgArgs A;
    memset(&A, 0, sizeof(A));
    A.plan = static_cast<const WgPlan*>(d_wplan);
    A.act = d_act_stash;
    A.grd = d_grad_stash;
    A.act_rec = hp->act_slots * mult;
    A.grd_rec = hp->grd_slots * mult;
    A.act_lo = mult == 2 ? hp->act_slots * 1024 : 0;
    A.grd_lo = mult == 2 ? hp->grd_slots * 1024 : 0;
    int max_tiles = 1;
    for (int j = 0; j < hp->n_jobs; ++j)
        if (hp->job[j].n_ot + hp->job[j].n_it > max_tiles) max_tiles = hp->job[j].n_ot + hp->job[j].n_it;
    A.slot_bytes = WG_TSTRIDE * max_tiles * mult;
    A.n_seg = n_rays * ((n_samples + 31) / 32);
    A.g = *grads;
    A.scratch = d_scratch;
    // one workgroup per CU, dealt to the jobs in proportion to their streamed bytes
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    const int nj = hp->n_jobs;
    int total_cost = 0;
    for (int j = 0; j < nj; ++j) total_cost += hp->cost[j];
    const int budget = ncu < WG_MAX_WGS ? ncu : WG_MAX_WGS;     // one resident workgroup per CU: a second round only repeats the pipeline fill / drain
                                // (measured 1.06 / 1.14 / 1.23 / 1.32 ms for 1 / 2 / 3 / 4 workgroups per CU)
    // proportional shares rounded down, then the workgroups left over go one at a time to the job whose workgroups
    // carry the most bytes each (every CU gets a workgroup and the slowest job sets the kernel's time)
    int n_wg[WG_MAX_JOBS], used = 0;
    for (int j = 0; j < nj; ++j) {
        int n = (int)((long long)budget * hp->cost[j] / total_cost);
        if (n < 1) n = 1;
        if (n > A.n_seg) n = A.n_seg;
        n_wg[j] = n;
        used += n;
    }
    while (used < budget) {
        int best = -1;
        for (int j = 0; j < nj; ++j)
            if (n_wg[j] < A.n_seg && (best < 0 || (long long)hp->cost[j] * n_wg[best] > (long long)hp->cost[best] * n_wg[j])) best = j;
        if (best < 0) break;
        n_wg[best]++;
        used++;
    }
    int acc_wg = 0;
    for (int j = 0; j < nj; ++j) {
        A.wg_start[j] = acc_wg;
        acc_wg += n_wg[j];
    }
    A.wg_start[nj] = acc_wg;
    // partial sums: area of every job's parts behind the G scratch, and the reduction's blocks (one per accumulator tile)
    A.partial = d_scratch + NFL_W * NFL_W;
    int part_floats = 0, red_blocks = 0;
    for (int j = 0; j < nj; ++j) {
        const int pw = hp->cost[j], nitw = (hp->job[j].n_it + hp->job[j].n_wi - 1) / hp->job[j].n_wi;
        // the instantiation nfl_wgrad_kernel picks for (pw, nitw): its NITW is the tile count a part stores
        const int inst = pw <= 4 ? (nitw <= 1 ? 1 : 2) : (pw <= 5 ? 2 : (pw <= 6 ? 4 : (nitw <= 5 ? 5 : (nitw <= 6 ? 6 : 8))));
        A.part_nitw[j] = inst;
        A.part_len[j] = 4 * WG_NOT * inst * 1024 + 4 * WG_NOT * 64;
        A.part_off[j] = part_floats;
        part_floats += n_wg[j] * A.part_len[j];
        A.red_start[j] = red_blocks;
        red_blocks += 4 * WG_NOT * inst;
    }
    A.red_start[nj] = red_blocks;
    if ((size_t)part_floats > (size_t)WG_MAX_WGS * (4 * WG_NOT * 8 * 1024 + 4 * WG_NOT * 64)) return NFL_EINVAL;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&nfl_wgrad_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 2 * WG_SLOT) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(&nfl_wgrad_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 4 * WG_SLOT) != hipSuccess)
            return NFL_ENODEV;
        attr_set = true;
    }
    if (2 * A.slot_bytes > 4 * WG_SLOT) return NFL_EINVAL;
    WgTensors T;
    for (int L = 0; L < NFL_NUM_LAYERS; ++L) {
        T.ptr[L] = grads->weight[L];
        T.n[L] = hp->w_numel[L];
        T.ptr[NFL_NUM_LAYERS + L] = grads->bias[L];
        T.n[NFL_NUM_LAYERS + L] = hp->b_numel[L];
    }
    T.ptr[2 * NFL_NUM_LAYERS] = d_scratch;
    T.n[2 * NFL_NUM_LAYERS] = NFL_W * NFL_W;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(nfl_wgrad_scale_kernel, dim3(16, WG_NTENS), dim3(256), 0, s, T, 0, d_gmax);
    if (n_rays == 0) return hipGetLastError() == hipSuccess ? NFL_OK : NFL_ELAUNCH;
#ifdef NFL_DIAG_WGRAD_PASSES
#error "NFL_DIAG_WGRAD_PASSES accumulated several launches with the atomic flush; it went with it (last: commit c170289)"
    // diagnostic (nfl_diag.h): the split stashes read by the ONE-product GEMM, hi images only (1) or d_hi + d_lo (2)
    {
        WgArgs B = A;
        B.act_lo = B.grd_lo = 0;
        B.slot_bytes = A.slot_bytes / mult;
        hipLaunchKernelGGL(nfl_wgrad_kernel<false>, dim3(acc_wg), dim3(256), 2 * B.slot_bytes, s, B);
        if (mult == 2 && NFL_DIAG_WGRAD_PASSES >= 2) {
            B.grd = d_grad_stash + (size_t)hp->grd_slots * 1024;
            hipLaunchKernelGGL(nfl_wgrad_kernel<false>, dim3(acc_wg), dim3(256), 2 * B.slot_bytes, s, B);
        }
    }
#else
    // NFL_PREC_F16X3: dW = sum_s (d_hi + d_lo) (x) (h_hi + h_lo) without the lo x lo term, in ONE pass over the hi and lo
    // records (fp16 x fp16 products are exact in the fp32 accumulators, so what is left is the 2^-22 lo x lo term and the
    // summation order); the bias gradients are sum_s (d_hi + d_lo).
    if (mult == 2) hipLaunchKernelGGL(nfl_wgrad_kernel<true>, dim3(acc_wg), dim3(256), 2 * A.slot_bytes, s, A);
    else hipLaunchKernelGGL(nfl_wgrad_kernel<false>, dim3(acc_wg), dim3(256), 2 * A.slot_bytes, s, A);
#endif
    // parts -> gradient tensors and G, summed in a fixed order and divided by the loss scale (what the tensors' zeroing above
    // still covers: elements no job owns, i.e. the heads a call leaves out)
    hipLaunchKernelGGL(nfl_wgrad_reduce_kernel, dim3(red_blocks), dim3(256), 0, s, A, d_gmax);

    // the composition through xyz_encoding_final (file header); G, db_dir, db_t0 are final (unscaled) by now
    const int W = NFL_W, H = NFL_W / 2;
    const int ld_dir = hp->w_numel[NFL_P_DIR] / H, ld_t0 = ut ? hp->w_numel[NFL_P_T0] / H : 0;
    const float* Wd = params->weight[NFL_P_DIR];
    const float* Wt = ut ? params->weight[NFL_P_T0] : nullptr;
    const float* Wf = params->weight[NFL_P_FINAL];
    const float* G = d_scratch;
    const float* Gt = d_scratch + (size_t)H * W;
    WgCompose Cc;
    memset(&Cc, 0, sizeof(Cc));
    auto add = [&](WgGemm g) {
        g.avec = g.bvec = 1;                 // G and W_fin rows are 1 KiB: checked below
        g.tiles_n = (g.N + 31) / 32;
        g.tile0 = Cc.n_wg;
        Cc.n_wg += g.tiles_n * ((g.M + 31) / 32);
        Cc.t[Cc.n_tasks++] = g;
    };
    WgGemm g;
    if (grads->weight[NFL_P_FINAL]) {       // dW_fin[i, j] = sum_r Wd[r, i] G[r, j] (+ sum_r Wt[r, i] Gt[r, j])
        memset(&g, 0, sizeof(g));
        g.A = Wd; g.sam = 1; g.sak = ld_dir; g.B = G; g.sbk = W; g.sbn = 1; g.K = H;
        if (ut) { g.A2 = Wt; g.sam2 = 1; g.sak2 = ld_t0; g.B2 = Gt; g.sbk2 = W; g.sbn2 = 1; g.K2 = H; }
        g.Cp = grads->weight[NFL_P_FINAL]; g.ldc = W; g.M = W; g.N = W;
        add(g);
    }
    if (grads->bias[NFL_P_FINAL]) {         // db_fin[i] = sum_r Wd[r, i] db_dir[r] (+ sum_r Wt[r, i] db_t0[r]): a 256 x 1 product
        memset(&g, 0, sizeof(g));
        g.A = Wd; g.sam = 1; g.sak = ld_dir; g.B = grads->bias[NFL_P_DIR]; g.sbk = 1; g.sbn = 1; g.K = H;
        if (ut) { g.A2 = Wt; g.sam2 = 1; g.sak2 = ld_t0; g.B2 = grads->bias[NFL_P_T0]; g.sbk2 = 1; g.sbn2 = 1; g.K2 = H; }
        g.Cp = grads->bias[NFL_P_FINAL]; g.ldc = 1; g.M = W; g.N = 1;
        add(g);
    }
    if (grads->weight[NFL_P_DIR]) {         // dW_dir[r, i] = sum_j G[r, j] W_fin[i, j] + db_dir[r] b_fin[i], i < 256
        memset(&g, 0, sizeof(g));
        g.A = G; g.sam = W; g.sak = 1; g.B = Wf; g.sbk = 1; g.sbn = W; g.K = W;
        g.u = grads->bias[NFL_P_DIR]; g.v = params->bias[NFL_P_FINAL];
        g.Cp = grads->weight[NFL_P_DIR]; g.ldc = ld_dir; g.M = H; g.N = W;
        add(g);
    }
    if (ut && grads->weight[NFL_P_T0]) {
        memset(&g, 0, sizeof(g));
        g.A = Gt; g.sam = W; g.sak = 1; g.B = Wf; g.sbk = 1; g.sbn = W; g.K = W;
        g.u = grads->bias[NFL_P_T0]; g.v = params->bias[NFL_P_FINAL];
        g.Cp = grads->weight[NFL_P_T0]; g.ldc = ld_t0; g.M = H; g.N = W;
        add(g);
    }
    if (((uintptr_t)d_scratch | (uintptr_t)Wf) & 15) return NFL_EINVAL;      // float4 loads along k (G, W_fin rows)
    if (Cc.n_wg > 0) hipLaunchKernelGGL(nfl_wgrad_compose_kernel, dim3(Cc.n_wg), dim3(256), 0, s, Cc);
    return hipGetLastError() == hipSuccess ? NFL_OK : NFL_ELAUNCH;
}

// Fold xyz_encoding_final into the layers that read its output, for the PACKED streams only (the parameters stay what
// they are): W_dir' = [W_dir[:, :256] W_fin | W_dir[:, 256:]], b_dir' = b_dir + W_dir[:, :256] b_fin, likewise for
// transient_encoding.0.  The caller hands in copies of W_dir / W_t0 (the side columns are kept, the first 256 columns
// overwritten) and receives the folded biases; nfl_pack_field(s) is then given these in place of the originals, and
// the streams carry no tiles for xyz_encoding_final (nfl_plan.cpp): one 256 x 256 layer less in the forward and in dgrad.
extern "C" int nfl_compose_forward(const nfl_field_params* params, int32_t has_t, int32_t n_side, int32_t n_tau,
                                   float* d_wdir_c, float* d_bdir_c, float* d_wt0_c, float* d_bt0_c, void* stream) {
    if (!params || !d_wdir_c || !d_bdir_c || (has_t && (!d_wt0_c || !d_bt0_c))) return NFL_EINVAL;
    const float* Wf = params->weight[NFL_P_FINAL];
    const float* bf = params->bias[NFL_P_FINAL];
    if (!Wf || !bf || !params->weight[NFL_P_DIR] || !params->bias[NFL_P_DIR]) return NFL_EINVAL;
    if (has_t && (!params->weight[NFL_P_T0] || !params->bias[NFL_P_T0])) return NFL_EINVAL;
    const int W = NFL_W, H = NFL_W / 2;
    WgCompose Cc;
    memset(&Cc, 0, sizeof(Cc));
    auto add = [&](WgGemm g) {
        g.tiles_n = (g.N + 31) / 32;
        g.tile0 = Cc.n_wg;
        Cc.n_wg += g.tiles_n * ((g.M + 31) / 32);
        Cc.t[Cc.n_tasks++] = g;
    };
    for (int which = 0; which < (has_t ? 2 : 1); ++which) {
        const int L = which ? NFL_P_T0 : NFL_P_DIR;
        const int ld = W + (which ? n_tau : n_side);
        const float* Ws = params->weight[L];
        WgGemm g;
        memset(&g, 0, sizeof(g));          // W'[r, i] = sum_j W[r, j] W_fin[j, i]   (rows of W are not 16-byte aligned: scalar loads)
        g.A = Ws; g.sam = ld; g.sak = 1; g.B = Wf; g.sbk = W; g.sbn = 1; g.K = W;
        g.Cp = which ? d_wt0_c : d_wdir_c; g.ldc = ld; g.M = H; g.N = W;
        add(g);
        memset(&g, 0, sizeof(g));          // b'[r] = b[r] + sum_j W[r, j] b_fin[j]
        g.A = Ws; g.sam = ld; g.sak = 1; g.B = bf; g.sbk = 1; g.sbn = 1; g.K = W;
        g.addm = params->bias[L];
        g.Cp = which ? d_bt0_c : d_bdir_c; g.ldc = 1; g.M = H; g.N = 1;
        add(g);
    }
    hipLaunchKernelGGL(nfl_wgrad_compose_kernel, dim3(Cc.n_wg), dim3(256), 0, static_cast<hipStream_t>(stream), Cc);
    return hipGetLastError() == hipSuccess ? NFL_OK : NFL_ELAUNCH;
}
