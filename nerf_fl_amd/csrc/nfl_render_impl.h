// nfl_render_impl.h -- the fused per-ray-chunk forward kernel.
//
// Replaces, for one pass (coarse or fine) of the reference's render_rays:
//   models/rendering.py:243-261  depth generation + o + d*z
//   models/nerf.py:19-32         positional encoding (in registers, never stored)
//   models/rendering.py:98-139   the point-chunk loop and all repeat/cat glue
//   models/nerf.py:153-212       the 8x256 MLP + heads   (MFMA, fp16 operands, fp32 accumulate)
//   models/rendering.py:141-226  alpha compositing       (wave scans, fp32)
//
// Mapping onto CDNA4
//   * workgroup = 4 waves, one per SIMD, up to 512 VGPR/AGPR each.  A workgroup owns a
//     contiguous range of rays and walks their samples in "segments" of 32 samples
//     (= one MFMA column block); a wave carries NCB segments at a time.
//   * the MLP is evaluated transposed, H^T[out,sample] = W[out,in] . H^T[in,sample]
//     with v_mfma_f32_32x32x16_f16.  Activations live in registers for the whole
//     network: the 32x32 fp32 accumulator tile of one layer, converted to fp16, IS the
//     B operand of the next layer (column = sample stays on the lane; the k-slot
//     permutation this implies is folded into the packed weights, nfl_plan.h).
//   * W streams global(L2) -> LDS through a 3-slot ring with global_load_lds (16 B per
//     lane, lane-linear = exactly the fragment image), one raw s_barrier per chunk and
//     a counted vmcnt so two chunks stay in flight across barriers; all four waves read
//     every fragment with conflict-free ds_read_b128.
//   * NSPLIT == 3: operands are split hi+lo in fp16 and three products are accumulated
//     (w_lo*x_hi + w_hi*x_lo + w_hi*x_hi): ~2^-21 relative error per product instead of
//     2^-11, at 3x the MFMA issue.  NSPLIT == 1 is the fast mode.
//   * compositing: per segment, an exclusive product scan of (1-alpha) over 32 lanes
//     gives the local transmittance; partial sums are linear in the incoming
//     transmittance, so segments (and tiles) of one ray are folded through a tiny LDS
//     record.  Nothing per-sample except the API's own (R,N) outputs is written.
#pragma once
#include <hip/hip_runtime.h>
#include <string.h>

#include <type_traits>

#include "../../include/nerf_fl_amd.h"
#include "nfl_diag.h"
#include "nfl_plan.h"
#include "nfl_prods.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <class V8> struct nfl_elem;
template <> struct nfl_elem<h8> { using type = _Float16; };
template <> struct nfl_elem<b8> { using type = __bf16; };
__device__ __forceinline__ f16v nfl_mfma(h8 a, h8 b, f16v c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f16v nfl_mfma(b8 a, b8 b, f16v c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
typedef float f4v __attribute__((ext_vector_type(4)));

#define NFL_DEV __device__ __forceinline__
// stash stores are streaming (nt): A/B on one box, training forward 1.69 ms with nt, 1.74 with plain stores (step 5.20 / 5.36 ms)
#define NFL_STREAM_STORE(v, p) __builtin_nontemporal_store(v, p)

// compile-time loop: f(integral_constant<int, I>) for I in [I0, I1)
template <int I0, int I1, class F>
NFL_DEV void nfl_static_for(F&& f) {
    if constexpr (I0 < I1) {
        f(std::integral_constant<int, I0>{});
        nfl_static_for<I0 + 1, I1>(f);
    }
}
#define NFL_NST 20            // floats in a segment / ray compositing record
#define NFL_REC 32            // record stride (floats)

// ray of pixel p of a frame (reference datasets/ray_utils.py:5-55); the ONE implementation behind nfl_gen_rays and the
// render kernel's camera prologue, so that both produce the same bits
template <class Cam>
NFL_DEV void nfl_cam_ray(const Cam& c, long long p, f4v& r0, f4v& r1) {
    const float i = (float)(p % c.width), j = (float)(p / c.width);
    const float dx = (i - c.cx) / c.fx, dy = -(j - c.cy) / c.fy, dz = -1.f;
    float d[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) d[r] = dx * c.c2w[4 * r] + dy * c.c2w[4 * r + 1] + dz * c.c2w[4 * r + 2];
    const float n = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    r0 = f4v{c.c2w[3], c.c2w[7], c.c2w[11], d[0] / n};
    r1 = f4v{d[1] / n, d[2] / n, c.near, c.far};
}

struct RenderArgs {
    const NflPlan* plan;      // device copy of the plan
    const char* packed;       // fragment stream, then bias table at plan->bias_off
    nfl_pass_args a;
    int n_chunks;             // chunks per tile consumed by this pass (prefix of the stream)
    int n_rt;                 // row tiles consumed by this pass
    int bias_off;
    int has_a;                // field has the appearance input
    int use_t;                // transient head evaluated in this pass
    int spr;                  // 32-sample segments per ray
    int rays_per_wg;
    float beta_min;
    int n_points, emb_stride;   // NFL_MODE_EMBED: rows / row stride (floats) of a.d_embedded
    int nfx_rt, ndir_rt;        // the field's frequency counts (<= the instantiation's: nfl_plan.h, "Encoder widths")
    int cx, cd;                 // 6 nfx_rt + 3, 6 ndir_rt + 3: widths of the encoded position / direction
    int n_a, n_tau;             // widths of the appearance / transient codes (<= 48 / 16; narrower: the k-steps are zero-padded)
    int gen_rays;               // rays come from `cam` (nfl_pass_args::h_cam), not from a.d_rays
    nfl_camera cam;
};

// ---------------------------------------------------------------------------------
// small math
// ---------------------------------------------------------------------------------
NFL_DEV float nfl_softplus(float x) { return x > 20.f ? x : log1pf(expf(x)); }   // torch default beta=1, threshold=20
NFL_DEV float nfl_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

// sin(2*pi*r) for r in about [-1, 2]; abs error < 2e-7 (minimax odd polynomial on [-1/4,1/4])
NFL_DEV float nfl_sin_rev(float r) {
    r = r - rintf(r);                                   // [-1/2, 1/2]
    float a = fabsf(r);
    a = a > 0.25f ? 0.5f - a : a;                       // sin(pi - t) = sin(t)
    a = copysignf(a, r);
    const float a2 = a * a;
    float p = 3.953670604e+01f;
    p = __builtin_fmaf(p, a2, -7.654978229e+01f);
    p = __builtin_fmaf(p, a2, 8.160100407e+01f);
    p = __builtin_fmaf(p, a2, -4.134165503e+01f);
    p = __builtin_fmaf(p, a2, 6.283185160e+00f);
    return a * p;
}

// x / (2*pi) as an unevaluated sum th + tl (exact to ~2^-45 relative)
NFL_DEV void nfl_turns(float x, float& th, float& tl) {
    const float C_HI = 0.15915493667125702f;            // fl32(1/(2 pi))
    const float C_LO = 6.4206382432985265e-09f;         // 1/(2 pi) - C_HI
    th = x * C_HI;
    const float e = __builtin_fmaf(x, C_HI, -th);
    tl = __builtin_fmaf(x, C_LO, e);
}

// feature f of [x | sin(2^0 x) | cos(2^0 x) | sin(2^1 x) ...] (3 columns per block);
// f, N compile-time after unrolling, coordinates as turns (th, tl) + raw value
// `pw` = per-frequency weights (LDS, broadcast reads): all ones, or the BARF coarse-to-fine weights
// of reference models/nerf.py:47-75 (computed on the host exactly as the reference does)
template <int N>
NFL_DEV float nfl_pe_feature(int f, const float (&raw)[3], const float (&th)[3], const float (&tl)[3], const float* pw) {
    if (f < 3) return raw[f];
    if (f >= 6 * N + 3) return 0.f;
    const int g = f - 3, k = g / 6, rem = g % 6, t = rem / 3, c = rem % 3;
    const float sc = (float)(1 << k);
    float r = __builtin_amdgcn_fractf(th[c] * sc) + tl[c] * sc;      // 2^k scaling is exact
    if (t) r += 0.25f;                                                // cos(y) = sin(y + pi/2)
    return pw[k] * nfl_sin_rev(r);
}

// relu on the bit pattern: one v_max_i32, and unlike v_max_f32 / v_med3_f32 it needs no canonicalising
// v_max_f32 x,x in front (negative floats are negative integers; -0 and negative NaNs become +0)
NFL_DEV float nfl_relu(float x) {
    int b = __builtin_bit_cast(int, x);
    b = b > 0 ? b : 0;
    return __builtin_bit_cast(float, b);
}
template <class E>
NFL_DEV unsigned nfl_pack2(float a, float b) {      // v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32 (round to nearest even)
    typedef E e2v __attribute__((ext_vector_type(2)));
    e2v r;
    r[0] = (E)a;
    r[1] = (E)b;
    return __builtin_bit_cast(unsigned, r);
}
// (x0, x1) -> packed 16-bit hi pair (returned) and the fp32 residuals x - float(hi): one pack + (fp16) two
// v_fma_mix_f32 reading the 16-bit halves directly (exact: a single rounding of x - hi, as the subtraction
// it replaces; no v_cvt_f32_f16 / v_pk_add_f32 + s_nop)
template <class E>
NFL_DEV unsigned nfl_split_pair(float x0, float x1, float& l0, float& l1) {
    const unsigned hi = nfl_pack2<E>(x0, x1);
    if constexpr (__is_same(E, _Float16)) {
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(hi), "v"(x0));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(hi), "v"(x1));
    } else {
        // bf16 -> f32 is a shift / a mask (gfx950 has no v_fma_mix_f32_bf16); the empty asm keeps the two
        // subtractions scalar (v_pk_add_f32 beside MFMAs costs more than two v_sub_f32)
        l0 = x0 - __builtin_bit_cast(float, hi << 16);
        asm volatile("" : "+v"(l0));
        l1 = x1 - __builtin_bit_cast(float, hi & 0xffff0000u);
    }
    return hi;
}

template <int NP, class V8>
NFL_DEV void nfl_split8(const float (&v)[8], V8 (&dst)[NP]) {
    using E = typename nfl_elem<V8>::type;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        if constexpr (NP == 2) {
            float l0, l1;
            reinterpret_cast<unsigned(&)[4]>(dst[0])[j / 2] = nfl_split_pair<E>(v[j], v[j + 1], l0, l1);
            reinterpret_cast<unsigned(&)[4]>(dst[NP - 1])[j / 2] = nfl_pack2<E>(l0, l1);
        } else {
            reinterpret_cast<unsigned(&)[4]>(dst[0])[j / 2] = nfl_pack2<E>(v[j], v[j + 1]);
        }
    }
}

// 8 values -> fp16 -> this lane's 16 B of a stash k-step (dst already includes lane*16); LO != 0: the fp16 residuals
// x - fp16(x) go LO bytes behind (split stashes of the three-product backward, nfl_plan.h)
template <int LO = 0>
NFL_DEV void nfl_stash8(const float (&v)[8], char* dst) {
    h8 t, tl;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        if constexpr (LO != 0) {
            float l0, l1;
            reinterpret_cast<unsigned(&)[4]>(t)[j / 2] = nfl_split_pair<_Float16>(v[j], v[j + 1], l0, l1);
            reinterpret_cast<unsigned(&)[4]>(tl)[j / 2] = nfl_pack2<_Float16>(l0, l1);
        } else {
            reinterpret_cast<unsigned(&)[4]>(t)[j / 2] = nfl_pack2<_Float16>(v[j], v[j + 1]);
        }
    }
    NFL_STREAM_STORE(t, reinterpret_cast<h8*>(dst));
    if constexpr (LO != 0) NFL_STREAM_STORE(tl, reinterpret_cast<h8*>(dst + LO));
}

// natural-order B operand of one k-step of a positional encoding: lane half h holds
// features 16*ks + 8*h + j.  Both candidates are evaluated per-lane via selects so the
// instruction stream is uniform.
template <int N, int NP, int LO = 0>
NFL_DEV void nfl_pe_kstep(int ks, int h, const float (&raw)[3], const float (&th)[3], const float (&tl)[3],
                          const float* pw, h8 (&dst)[NP], char* stash = nullptr) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int f0 = 16 * ks + j, f1 = f0 + 8;
        // select the feature descriptor by lane half, then evaluate once
        if (f0 < 3 || f0 >= 6 * N + 3 || f1 >= 6 * N + 3) {
            const float v0 = nfl_pe_feature<N>(f0, raw, th, tl, pw);
            const float v1 = nfl_pe_feature<N>(f1, raw, th, tl, pw);
            v[j] = h ? v1 : v0;
        } else {
            const int g0 = f0 - 3, g1 = f1 - 3;
            const int k0 = g0 / 6, k1 = g1 / 6, t0 = (g0 % 6) / 3, t1 = (g1 % 6) / 3, c0 = g0 % 3, c1 = g1 % 3;
            const float sc = h ? (float)(1 << k1) : (float)(1 << k0);
            const float thc = h ? th[c1] : th[c0];
            const float tlc = h ? tl[c1] : tl[c0];
            const float ph = h ? 0.25f * t1 : 0.25f * t0;
            const float r = __builtin_amdgcn_fractf(thc * sc) + tlc * sc + ph;
            v[j] = (h ? pw[k1] : pw[k0]) * nfl_sin_rev(r);
        }
    }
    nfl_split8<NP>(v, dst);
    if (stash) {        // the operand images ARE the stash (hi; with LO the residuals too)
        NFL_STREAM_STORE(dst[0], reinterpret_cast<h8*>(stash));
        if constexpr (LO != 0 && NP == 2) NFL_STREAM_STORE(dst[NP - 1], reinterpret_cast<h8*>(stash + LO));
    }
}

// ---------------------------------------------------------------------------------
// weight ring: global -> LDS by LDS-DMA, 3 slots, prefetch distance 2
// ---------------------------------------------------------------------------------
template <int SLOT_BYTES, int MAXP_>
struct NflRing {
    static constexpr int MAXP = MAXP_;     // DMA pieces (1 KiB per wave-instruction) every wave issues per chunk
    const char* gsrc;
    const int* chunk_off;
    char* lds;          // ring base (LDS)
    int n_chunks;
    int c_issue;        // next chunk (index within the per-tile stream) to issue
    int s_issue;        // slot it goes to
    int s_read;         // slot of the next chunk to consume
    int wave, lane;
    // chunk currently being issued (pieces are spread over the MFMA loop of the chunk being consumed)
    const char* i_src;
    char* i_dst;
    int i_nbytes;

#ifdef NFL_STAMPS
    unsigned long long t_wait = 0, t_bar = 0;   // cycles in consume(): DMA wait / workgroup barrier
#endif
    int n_off0, n_off1;   // table entries of chunk c_issue, fetched one step ahead (no LDS latency after the barrier)
    // Not ring state, but it travels with the ring through every layer: per-lane running maximum (packed u16 pair) of
    // the |fp16 bit patterns| the activation epilogues have formed.  >= 0x7c00 at the end of the kernel means an
    // activation left fp16's range (the conversion gave inf); reported through nfl_pass_args::d_status.
    unsigned ovf = 0;

    NFL_DEV void begin_issue() {
        i_nbytes = n_off1 - n_off0;
        i_src = gsrc + n_off0;               // wave-uniform; the lane offset is added per piece (keeps no 64-bit VGPR live)
        i_dst = lds + s_issue * SLOT_BYTES;
        c_issue = c_issue + 1 == n_chunks ? 0 : c_issue + 1;
        s_issue = s_issue == 2 ? 0 : s_issue + 1;
        n_off0 = __builtin_amdgcn_readfirstlane(chunk_off[c_issue]);
        n_off1 = __builtin_amdgcn_readfirstlane(chunk_off[c_issue + 1]);
    }
    template <int P>
    NFL_DEV void piece() {
        if constexpr (P < MAXP) {
            // uniform byte offset (SALU min), one VALU add for the lane: SGPR base + 32-bit VGPR offset
            unsigned byte = (unsigned)(wave + 4 * P) * 1024u;
            const unsigned last = (unsigned)i_nbytes - 1024u;
            byte = byte < last ? byte : last;                      // surplus pieces re-copy the last KiB
            const unsigned vo = byte + (threadIdx.x & 63) * 16u;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(i_src + vo),
                (__attribute__((address_space(3))) void*)(i_dst + byte), 16, 0, 0);
        }
    }
    template <int P0, int P1>
    NFL_DEV void pieces() {                 // pieces [P0, P1)
        nfl_static_for<P0, P1>([&](auto P) __attribute__((always_inline)) { piece<decltype(P)::value>(); });
    }
    NFL_DEV void prime() {
        n_off0 = __builtin_amdgcn_readfirstlane(chunk_off[c_issue]);
        n_off1 = __builtin_amdgcn_readfirstlane(chunk_off[c_issue + 1]);
        begin_issue();
        pieces<0, MAXP>();
        begin_issue();
        pieces<0, MAXP>();
    }
    // Wait for the oldest chunk in flight, make it visible to all waves and return this lane's
    // read base; the caller then issues the MAXP pieces of the next chunk (piece<P>()) while it
    // computes, into the slot everybody has just finished reading.
    // EXTRA: VMEM ops (stash stores) known to have been issued after the pieces of the chunk waited for,
    // besides the MAXP pieces of the next one -- without it the wait would also sit on those stores.
    template <int EXTRA = 0>
    NFL_DEV const char* consume() {
#if defined(NFL_STAMPS) && NFL_STAMPS >= 2
        const unsigned long long c0 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(MAXP + EXTRA) : "memory");
        const unsigned long long c1 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_barrier();
        t_wait += c1 - c0;
        t_bar += __builtin_amdgcn_s_memtime() - c1;
#else
        // all but the MAXP (+EXTRA) youngest VMEM ops (= the younger chunk's pieces) are done
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(MAXP + EXTRA) : "memory");
        __builtin_amdgcn_s_barrier();
#endif
        asm volatile("" ::: "memory");
        begin_issue();
        const char* base = lds + s_read * SLOT_BYTES + (threadIdx.x & 63) * 16;
        s_read = s_read == 2 ? 0 : s_read + 1;
        return base;
    }
};

// ---------------------------------------------------------------------------------
// MFMA building blocks
// ---------------------------------------------------------------------------------
template <int NP, int NCB>
NFL_DEV void nfl_bias_init(f16v (&acc)[NCB], const float* bias_rt, int h) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f4v b = *reinterpret_cast<const f4v*>(bias_rt + 8 * q + 4 * h);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            acc[cb][4 * q + 0] = b[0];
            acc[cb][4 * q + 1] = b[1];
            acc[cb][4 * q + 2] = b[2];
            acc[cb][4 * q + 3] = b[3];
        }
    }
}

// One row tile, software-pipelined in source order.  Every MFMA is followed by a few
// "fillers" that fit in the issue slots it leaves free (an MFMA holds the issue port for 8 of
// its 32 cycles): the LDS reads of k-step k+2, one LDS-DMA piece of the chunk being prefetched,
// and a slice of the PREVIOUS tile's VALU epilogue.  sched_barrier(0) after each micro-slice
// pins that order (hipcc otherwise emits the DMA pieces and the epilogue back to back after
// the barrier, with the matrix pipe idle).
//   getb(K, cb, part) -> B operand of k-step K;  epi.template step<K, NK>() runs the epilogue
//   work assigned to k-step K;  pieces P0+k are issued at k-step k.
// The weight fragments are read with hand-issued ds_read_b128 and hand-counted s_waitcnt lgkmcnt(N): left to
// hipcc, every third k-step got an `s_waitcnt lgkmcnt(0)` that also waits for the reads issued one instruction
// earlier for k+2, so the full LDS latency was exposed once per three k-steps (1.4x the MFMA time with three
// products per k-step, 2x with one).  LDS operations return in order, so "all but the N youngest" is exact: N =
// the reads of k-step k+1.  Any LDS operation the compiler adds in between only makes the wait stricter.
typedef unsigned nfl_u4 __attribute__((ext_vector_type(4)));
template <int OFF>
NFL_DEV nfl_u4 nfl_lds_read128(unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field is 16 bits");
    nfl_u4 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
// wait until at most N LDS operations are outstanding; the operands ride through so that their users stay below
template <int N, int NREAD, int NWP>
NFL_DEV void nfl_lds_wait(nfl_u4 (&w)[NWP]) {
    if constexpr (NREAD == 2) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(w[0]), "+v"(w[1]) : "n"(N));
    else asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(w[0]) : "n"(N));
}

// DEPTH = how many k-steps ahead the fragments are read (DEPTH + 1 register sets).  Two k-steps are 6 MFMAs with
// three products per k-step but only 2 with one: the single-product kernels read 4 ahead, or every k-step waits
// out most of the LDS latency (in-kernel stamps: 3.3 k cycles per 32-MFMA row tile with DEPTH 2).
#ifndef NFL_DEPTH_X3
#define NFL_DEPTH_X3 2
#endif
// PRODS (three-product mode only): which of the two correction products a layer issues besides w_hi x_hi --
// bit 0: w_lo x_hi (the weights' fp16 residuals; without it the layer's weights are fp16-rounded and their lo fragments
// are not even read from LDS), bit 1: w_hi x_lo (the activations' residuals).  3 = the full f16x3 product.  The per-layer
// plan is NFL_PRODS (nfl_prods.h), chosen by measurement against the parity bar (tests/report_parity.py).
// bit 2 (NP == 1 only): the stream carries hi + lo WEIGHT fragments although the B operands are single fp16 images -- the
// default dgrad (nfl_dgrad.hip): W_hi d_hi + W_lo d_hi, the weights to fp32 class, the gradients fp16.
template <int PRODS, int NP, int NCB, int NK, int P0, class V8, class GetB, class Epi, class Ring,
          int DEPTH = ((NP == 1 && (PRODS & 4) == 0) ? 4 : NFL_DEPTH_X3)>
NFL_DEV void nfl_tile_p(f16v (&acc)[NCB], const char* wl, const int frag0, GetB&& getb, Epi&& epi, Ring& ring) {
    constexpr int NWP = (NP == 2 || (PRODS & 4) != 0) ? 2 : 1;          // weight fragments per k-step: hi (+ lo)
    constexpr int KSB = 1024 * NWP;
    constexpr int NW = DEPTH + 1;
    constexpr bool W_LO = NWP == 2 && (PRODS & 1) != 0, X_LO = NP == 2 && (PRODS & 2) != 0;
    constexpr int NREAD = W_LO ? 2 : 1;     // LDS reads per k-step
    (void)frag0;                          // == P0 (kept in the signature for the callers' readability)
    nfl_u4 w[NW][NWP];
    const unsigned wa = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)wl;
    auto load = [&](auto K) __attribute__((always_inline)) {
        constexpr int k = decltype(K)::value;
        w[k % NW][0] = nfl_lds_read128<(P0 + k) * KSB>(wa);
        if constexpr (W_LO) w[k % NW][NWP - 1] = nfl_lds_read128<(P0 + k) * KSB + 1024>(wa);
    };
    nfl_static_for<0, (DEPTH < NK ? DEPTH : NK)>([&](auto K) __attribute__((always_inline)) { load(K); });
    epi.early();                         // VALU work that hides the latency of the first LDS reads
    __builtin_amdgcn_sched_barrier(0);
    nfl_static_for<0, NK>([&](auto K) __attribute__((always_inline)) {
        constexpr int k = decltype(K)::value;
        // k-step k has landed; the reads of k+1 .. k+DEPTH-1 (already issued) may still be in flight
        constexpr int younger = (NK - 1 - k) < (DEPTH - 1) ? (NK - 1 - k) : (DEPTH - 1);
        nfl_lds_wait<younger * NREAD, NREAD>(w[k % NW]);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            // fillers of the k-step (the LDS reads of k + DEPTH, one DMA piece) ride behind its first two MFMAs
            if constexpr (W_LO) {
                acc[cb] = nfl_mfma(__builtin_bit_cast(V8, w[k % NW][NWP - 1]), getb(K, cb, 0), acc[cb]);
                if (cb == 0) {
                    if constexpr (k + DEPTH < NK) load(std::integral_constant<int, k + DEPTH>{});
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (X_LO) {
                acc[cb] = nfl_mfma(__builtin_bit_cast(V8, w[k % NW][0]), getb(K, cb, NP - 1), acc[cb]);
                if (cb == 0) {
                    if constexpr (!W_LO && k + DEPTH < NK) load(std::integral_constant<int, k + DEPTH>{});
                    ring.template piece<P0 + k>();
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            acc[cb] = nfl_mfma(__builtin_bit_cast(V8, w[k % NW][0]), getb(K, cb, 0), acc[cb]);
            if (cb == 0) {
                if constexpr (!W_LO && !X_LO && k + DEPTH < NK) load(std::integral_constant<int, k + DEPTH>{});
                if constexpr (!X_LO) ring.template piece<P0 + k>();
            }
            if (cb == NCB - 1) epi.template step<k, NK>();
            __builtin_amdgcn_sched_barrier(0);
        }
    });
}
template <int NP, int NCB, int NK, int P0, class V8, class GetB, class Epi, class Ring>
NFL_DEV void nfl_tile(f16v (&acc)[NCB], const char* wl, const int frag0, GetB&& getb, Epi&& epi, Ring& ring) {
    nfl_tile_p<3, NP, NCB, NK, P0, V8>(acc, wl, frag0, getb, epi, ring);
}

// max over the NFL_GMAX_SLOTS words the compositing backward left (bit patterns of non-negative floats order
// like unsigned integers); wave-uniform result
NFL_DEV unsigned nfl_gmax_bits(const float* d_gmax) {
    unsigned v = 0u;
    if (d_gmax)
        for (int i = threadIdx.x & 63; i < NFL_GMAX_SLOTS; i += 64) {
            const unsigned o = reinterpret_cast<const unsigned*>(d_gmax)[i];
            v = o > v ? o : v;
        }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned o = __shfl_xor(v, d);
        v = o > v ? o : v;
    }
    return __builtin_amdgcn_readfirstlane(v);
}

struct NflNoEpi {
    template <int K, int NK> NFL_DEV void step() {}
    NFL_DEV void early() {}
};
#ifndef NFL_EPI_EARLY
#define NFL_EPI_EARLY 2      // pair-ops done before the first MFMA of the following tile
#endif

// Epilogue of an accumulator tile -> the two k-steps (ks, ks+1) of the next layer's B operand
// (and, in the training forward, the fp16 activation stash), cut into 8 pair-ops per column
// block so it can be spread over the k-steps of the following tile.
template <int NP, int NCB, bool RELU, bool STASH, int NOUT, int MSLOT, int LO = 0>
struct NflActEpi {
    const f16v (&acc)[NCB];
    h8 (&out)[NOUT][NCB][NP];
    const int ks;
    char* const (&stash)[NCB];
    const int slot;
    char* const (&mstash)[NCB];      // relu-mask records of the lane's segments (training forward)
    const int mword;                 // mask word of this tile
    unsigned (&mq)[NCB][4];          // the words of the current group of four tiles: one dwordx4 store per group
    unsigned& ovf;                   // NflRing::ovf
    h8 tmp[NCB];
    h8 tmpl[LO != 0 ? NCB : 1];      // residual halves for the split stash (LO: their byte offset behind the hi image)
    unsigned m32[NCB];

    template <int OP>
    NFL_DEV void pair() {                      // OP 0..7: elements 2*OP, 2*OP+1 of the 16 accumulators
        constexpr int s = OP / 4, j = 2 * (OP % 4);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            float x0 = acc[cb][8 * s + j], x1 = acc[cb][8 * s + j + 1];
            if (RELU) {
                x0 = nfl_relu(x0);
                x1 = nfl_relu(x1);
            }
            unsigned hi;
            if constexpr (NP == 2) {
                float l0, l1;
                hi = nfl_split_pair<_Float16>(x0, x1, l0, l1);
                const unsigned lo = nfl_pack2<_Float16>(l0, l1);
                reinterpret_cast<unsigned(&)[4]>(out[ks + s][cb][NP - 1])[j / 2] = lo;
                if constexpr (STASH && LO != 0) {
                    reinterpret_cast<unsigned(&)[4]>(tmpl[cb])[j / 2] = lo;
                    if (OP % 4 == 3) NFL_STREAM_STORE(tmpl[cb], reinterpret_cast<h8*>(stash[cb] + LO + (slot + s) * 1024));
                }
            } else {
                hi = nfl_pack2<_Float16>(x0, x1);
            }
            reinterpret_cast<unsigned(&)[4]>(out[ks + s][cb][0])[j / 2] = hi;
            {   // range tracking: after relu the halves are non-negative, so their bit patterns order like the values
                const unsigned mag = RELU ? hi : (hi & 0x7fff7fffu);
                asm("v_pk_max_u16 %0, %0, %1" : "+v"(ovf) : "v"(mag));
            }
            if (STASH) {        // the fp16 hi operand IS the stashed activation
                reinterpret_cast<unsigned(&)[4]>(tmp[cb])[j / 2] = hi;
                if (OP % 4 == 3) NFL_STREAM_STORE(tmp[cb], reinterpret_cast<h8*>(stash[cb] + (slot + s) * 1024));
                if (RELU) {     // relu mask of the pair for the dgrad kernel: bit 2*OP / 16 + 2*OP (nfl_plan.h)
                    unsigned on;
                    asm("v_pk_min_u16 %0, %1, %2" : "=v"(on) : "v"(hi), "s"(0x00010001u));
                    m32[cb] = OP == 0 ? on : ((on << (2 * OP)) | m32[cb]);
                    if (OP == 7) {
                        // mask words are grouped by four tiles (mw0 is a multiple of 4 for every layer): lane l keeps
                        // words 4g..4g+3 in 16 contiguous bytes, record layout [group][lane][4]
                        mq[cb][MSLOT] = m32[cb];              // MSLOT = mword & 3, known at compile time
                        if (MSLOT == 3) {
                            typedef unsigned nfl_mq4 __attribute__((ext_vector_type(4)));
                            const nfl_mq4 v = {mq[cb][0], mq[cb][1], mq[cb][2], mq[cb][3]};
                            NFL_STREAM_STORE(v, reinterpret_cast<nfl_mq4*>(mstash[cb] + (mword >> 2) * 1024));
                        }
                    }
                }
            }
        }
    }
    template <int K, int NK>
    NFL_DEV void step() {                      // the remaining pair-ops, spread evenly over the k-steps
        constexpr int R = 8 - NFL_EPI_EARLY;
        nfl_static_for<NFL_EPI_EARLY + (R * K) / NK, NFL_EPI_EARLY + (R * (K + 1)) / NK>([&](auto O) __attribute__((always_inline)) {
            pair<decltype(O)::value>();
        });
    }
    NFL_DEV void early() {
        nfl_static_for<0, NFL_EPI_EARLY>([&](auto O) __attribute__((always_inline)) { pair<decltype(O)::value>(); });
    }
    NFL_DEV void all() {
        nfl_static_for<0, 8>([&](auto O) __attribute__((always_inline)) { pair<decltype(O)::value>(); });
    }
};

// A dense layer of NRT row tiles reading inA[ksA0..+NKA) then inB[ksB0..+NKB), TPC tiles per
// ring chunk.  The epilogue of tile i-1 rides in the MFMA shadows of tile i.
template <int NP, int NCB, int NKA, int NKB, bool RELU, int NRT, int TPC, bool STASH, int LO = 0, int PRODS = 3, int NINA, int NINB, int NOUT, class Ring>
NFL_DEV void nfl_dense(Ring& ring, const float* bias_lds, int& rt, int h,
                       const h8 (&inA)[NINA][NCB][NP], int ksA0,
                       const h8 (&inB)[NINB][NCB][NP], int ksB0,
                       h8 (&out)[NOUT][NCB][NP], int out_ks0, char* const (&stash)[NCB], int slot0,
                       char* const (&mstash)[NCB], int mw0) {
    constexpr int NK = NKA + NKB;
    constexpr int NST = 2 * NCB * (LO != 0 ? 2 : 1);     // activation-stash stores of one tile's epilogue (the mask
                                                         // words go out once per four tiles: not counted, the wait is only stricter)
    unsigned mq[NCB][4];
    f16v acc[2][NCB];
    const char* wl = nullptr;
    auto getb = [&](auto K, int cb, int part) __attribute__((always_inline)) -> const h8& {
        constexpr int k = decltype(K)::value;
        if constexpr (k < NKA) return inA[ksA0 + k][cb][part];
        else return inB[ksB0 + k - NKA][cb][part];
    };
    nfl_static_for<0, NRT>([&](auto I) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value;
        // tile i-1 carried the stash stores of tile i-2's epilogue, issued after this chunk's pieces
        if (i % TPC == 0) wl = ring.template consume<(STASH && TPC == 1 && i >= 2) ? NST : 0>();
        constexpr int frag0 = (i % TPC) * NK;
        nfl_bias_init<NP, NCB>(acc[i & 1], bias_lds + (rt + i) * 32, h);
        if constexpr (i > 0) {
            NflActEpi<NP, NCB, RELU, STASH, NOUT, (i - 1) & 3, LO> epi{acc[(i - 1) & 1], out, out_ks0 + 2 * (i - 1), stash, slot0 + 2 * (i - 1), mstash, mw0 + i - 1, mq, ring.ovf};
            nfl_tile_p<PRODS, NP, NCB, NK, frag0, h8>(acc[i & 1], wl, frag0, getb, epi, ring);
        } else {
            NflNoEpi epi;
            nfl_tile_p<PRODS, NP, NCB, NK, frag0, h8>(acc[i & 1], wl, frag0, getb, epi, ring);
        }
        // pieces the k-loop of this chunk did not get to
        if (i % TPC == TPC - 1 || i == NRT - 1) ring.template pieces<((i % TPC) + 1) * NK, Ring::MAXP>();
    });
    NflActEpi<NP, NCB, RELU, STASH, NOUT, (NRT - 1) & 3, LO> last{acc[(NRT - 1) & 1], out, out_ks0 + 2 * (NRT - 1), stash, slot0 + 2 * (NRT - 1), mstash, mw0 + NRT - 1, mq, ring.ovf};
    last.all();
    rt += NRT;
}

// a single head tile (own chunk); the caller interprets the accumulator rows
template <int NP, int NCB, int NK, int PRODS = 3, int NIN, class Ring>
NFL_DEV void nfl_head(Ring& ring, const float* bias_lds, int& rt, int h,
                      const h8 (&in)[NIN][NCB][NP], int ks0, f16v (&acc)[NCB]) {
    const char* wl = ring.consume();
    nfl_bias_init<NP, NCB>(acc, bias_lds + rt * 32, h);
    auto getb = [&](auto K, int cb, int part) __attribute__((always_inline)) -> const h8& {
        return in[ks0 + decltype(K)::value][cb][part];
    };
    NflNoEpi epi;
    nfl_tile_p<PRODS, NP, NCB, NK, 0, h8>(acc, wl, 0, getb, epi, ring);
    ring.template pieces<NK, Ring::MAXP>();
    rt += 1;
}

// ---------------------------------------------------------------------------------
// depths (reference models/rendering.py:243-259); every operation separately rounded
// ---------------------------------------------------------------------------------
template <class PA>
NFL_DEV float nfl_z_plain(const PA& a, float near, float far, int i) {
    const float s = a.d_lin[i];
    const float oms = 1.0f - s;
    if (!a.use_disp) return near * oms + far * s;
    return 1.0f / (1.0f / near * oms + 1.0f / far * s);
}
template <class PA>
NFL_DEV float nfl_z_at(const PA& a, int ray, float near, float far, int i) {
    const int N = a.n_samples;
    if (a.d_z) return a.d_z[(size_t)ray * N + i];
    float z = nfl_z_plain(a, near, far, i);
    if (a.perturb > 0.f) {
        const float zm = nfl_z_plain(a, near, far, i > 0 ? i - 1 : 0);
        const float zp = nfl_z_plain(a, near, far, i < N - 1 ? i + 1 : N - 1);
        const float upper = i < N - 1 ? 0.5f * (z + zp) : z;
        const float lower = i > 0 ? 0.5f * (zm + z) : z;
        const float pr = a.perturb * a.d_perturb_rand[(size_t)ray * N + i];
        z = lower + (upper - lower) * pr;
    }
    return z;
}

// 32-lane helpers (both halves of the wave run them independently)
NFL_DEV float nfl_sum32(float v) {
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m, 32);
    return v;
}

// ---------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------
template <int NSPLIT, int NCB, int NFX>
struct NflRenderCfg {
    static constexpr int NP = NSPLIT == 3 ? 2 : 1;
    static constexpr int NKP = (6 * NFX + 3 + 15) / 16;
    static constexpr int KSB = 1024 * NP;
    static constexpr int MAXKS = 16 + (NKP > 5 ? NKP : 5);
    static constexpr int SLOT = MAXKS * KSB;
    static constexpr int MAXP = (SLOT + 4095) / 4096;
    static constexpr int NSLOT = 4 * NCB;
    static constexpr int LDS_RING = 3 * SLOT;
    static constexpr int LDS_BIAS = NFL_MAX_RT * 32 * 4;
    static constexpr int LDS_REC = (NSLOT + 2) * NFL_REC * 4;
    static constexpr int LDS_CHK = (NFL_MAX_CHUNKS + 8 + 32 + 8) * 4; // chunk offsets + 32 positional-encoding weights + 4 loss partials
    static constexpr int LDS_BYTES = LDS_RING + LDS_BIAS + LDS_REC + LDS_CHK;
};

// Diagnostic build only (make diag, -DNFL_STAMPS): per-phase s_memtime totals of every wave, written to a
// buffer nothing else reads.  No stamp code exists in the product build.
#ifdef NFL_STAMPS
#define NFL_NSTAMP 20
__device__ unsigned long long nfl_stamp_buf[1024 * 4 * NFL_NSTAMP];
// diagnostic build: copy the per-wave phase cycle totals of the last launch to the host
extern "C" int nfl_debug_stamps(unsigned long long* host, int n_entries) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(nfl_stamp_buf), sizeof(unsigned long long) * n_entries) == hipSuccess ? 0 : -1;
}
#define NFL_STAMP(i)                                                      \
    do {                                                                  \
        const unsigned long long t_now = __builtin_amdgcn_s_memtime();    \
        t_acc[i] += t_now - t_last;                                       \
        t_last = t_now;                                                   \
    } while (0)
#else
#define NFL_STAMP(i) do {} while (0)
#endif

#define NFL_LOSS_ON(a) ((a).d_loss_target != nullptr)
// The kernel's argument block, re-read from the kernarg segment.  Values loaded through the returned pointer cannot be
// hoisted above the call (the empty asm makes the pointer opaque), so arguments that are only needed in the cold parts
// of a tile (ray set-up, compositing, outputs, loss) are s_load'ed there instead of being kept in SGPRs -- or rather in
// SGPR spill lanes of VGPRs -- across the whole MLP, where every register is spoken for.
struct RenderArgs;
typedef const __attribute__((address_space(4))) RenderArgs* NflKArgs;
NFL_DEV NflKArgs nfl_kargs() {
    NflKArgs p = (NflKArgs)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}

// the camera of a ray-generating pass: in the kernarg segment (copied from nfl_pass_args::h_cam at launch) or, when the
// caller keeps it in device memory (d_cam: graph replays), behind that pointer -- read with scalar loads either way
typedef const __attribute__((address_space(4))) nfl_camera* NflKCam;
NFL_DEV NflKCam nfl_kcam(NflKArgs K) {
    return K->a.d_cam ? (NflKCam)(unsigned long long)K->a.d_cam : &K->cam;
}

#define NFL_MODE_RENDER 0
#define NFL_MODE_STASH 1      // render + fp16 activation stash for the backward
#define NFL_MODE_STASH2 3     // as STASH, with split (hi + lo) activation records for the three-product backward
#define NFL_MODE_EMBED 2      // NeRF.forward on already-encoded inputs (reference models/nerf.py:153-212): no
                              // depth generation / encoding / compositing, 32 points per segment
template <int NSPLIT, int NCB, int NFX, int MODE>
__global__ __launch_bounds__(256, 1) void nfl_render_kernel(const RenderArgs A) {
    constexpr bool STASH = MODE == NFL_MODE_STASH || MODE == NFL_MODE_STASH2, EMBED = MODE == NFL_MODE_EMBED;
    using C = NflRenderCfg<NSPLIT, NCB, NFX>;
    constexpr int NP = C::NP, NKP = C::NKP, NSLOT = C::NSLOT;
    constexpr int SMULT = MODE == NFL_MODE_STASH2 ? 2 : 1;                              // records per segment: hi (+ lo)
    constexpr int LO = MODE == NFL_MODE_STASH2 ? nfl_act_slots(NKP) * 1024 : 0;         // byte offset of the lo record
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // small tables first (so ds_read immediates reach them from one base register), ring last
    float* const bias_lds = reinterpret_cast<float*>(smem);
    float* const rec_lds = reinterpret_cast<float*>(smem + C::LDS_BIAS);   // [NSLOT] segment records, then carry[2]
    int* const chk_lds = reinterpret_cast<int*>(smem + C::LDS_BIAS + C::LDS_REC);
    float* const pw_lds = reinterpret_cast<float*>(chk_lds + NFL_MAX_CHUNKS + 8);   // [0,16): xyz freqs, [16,32): dir freqs
    float* const loss_lds = pw_lds + 32;                                            // [4] fused-loss partial sums of this workgroup

    const nfl_pass_args& a = A.a;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, c = lane & 31;
    const int N = a.n_samples, SPR = A.spr;

    const int ray0 = blockIdx.x * A.rays_per_wg;
    int ray1 = ray0 + A.rays_per_wg;
    if (ray1 > a.n_rays) ray1 = a.n_rays;
    if (ray0 >= ray1) return;
    const int seg_end = (ray1 - ray0) * SPR;
    const int ntiles = (seg_end + NSLOT - 1) / NSLOT;

    {   // bias table and chunk offsets -> LDS (LDS reads keep the VMEM queue free for the ring)
        const float* bg = reinterpret_cast<const float*>(A.packed + A.bias_off);
        for (int i = tid; i < A.n_rt * 32; i += 256) bias_lds[i] = bg[i];
        for (int i = tid; i <= A.n_chunks; i += 256) chk_lds[i] = A.plan->chunk_off[i];
        if (tid < 16) pw_lds[tid] = (a.d_pe_w_xyz && tid < A.nfx_rt) ? a.d_pe_w_xyz[tid] : 1.f;
        else if (tid < 32) pw_lds[tid] = (a.d_pe_w_dir && tid - 16 < A.ndir_rt) ? a.d_pe_w_dir[tid - 16] : 1.f;
        else if (tid < 36) loss_lds[tid - 32] = 0.f;
    }
    __syncthreads();

    NflRing<C::SLOT, C::MAXP> ring;
    ring.gsrc = A.packed;
    ring.chunk_off = chk_lds;
    ring.lds = smem + C::LDS_BIAS + C::LDS_REC + C::LDS_CHK;
    ring.n_chunks = A.n_chunks;
    ring.c_issue = 0;
    ring.s_issue = 0;
    ring.s_read = 0;
    ring.wave = wave;
    ring.lane = lane;
    ring.prime();

#ifdef NFL_STAMPS
    unsigned long long t_acc[NFL_NSTAMP] = {};
    unsigned long long t_last = __builtin_amdgcn_s_memtime();
#endif
    for (int tile = 0; tile < ntiles; ++tile) {
        NflKArgs K = nfl_kargs();          // ray set-up: arguments loaded here, not carried through the MLP
        // ------------------------------------------------------------ per-sample setup
        int s_ray[NCB], s_idx[NCB];
        bool s_ok[NCB];
        float s_z[NCB], s_dl[NCB];
        h8 P[NKP][NCB][NP];
        h8 X[16][NCB][NP], Y[16][NCB][NP];
        char* st[NCB];        // this lane's slice of the segment's activation record (training forward) or null
        char* mst[NCB];       // ... and of its relu-mask record
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const int g = tile * NSLOT + wave * NCB + cb;        // segment index inside this workgroup
            const bool seg_ok = g < seg_end;
            const int gg = seg_ok ? g : seg_end - 1;
            const int ray = ray0 + gg / SPR;
            const int i = (gg % SPR) * 32 + c;
            const bool ok = seg_ok && i < N;
            const int ii = i < N ? i : N - 1;
            if constexpr (EMBED) {
                // point b = 32 * segment + c of an (n_points, row) matrix [xyz enc | dir enc (+a) | tau]
                const int b = ray * 32 + c;
                const bool pok = seg_ok && b < K->n_points;
                const float* xr = K->a.d_embedded + (size_t)(b < K->n_points ? b : K->n_points - 1) * K->emb_stride;
                st[cb] = nullptr;
                mst[cb] = nullptr;
                s_ray[cb] = ray;
                s_idx[cb] = c;
                s_ok[cb] = pok;
                s_z[cb] = 0.f;
                s_dl[cb] = 0.f;
#pragma unroll
                for (int ks = 0; ks < NKP; ++ks) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int f = 16 * ks + 8 * h + j;
                        v[j] = f < K->cx ? xr[f] : 0.f;
                    }
                    nfl_split8<NP>(v, P[ks][cb]);
                }
                continue;
            }
            f4v r0, r1;
            if (K->gen_rays) {
                const NflKCam cam = nfl_kcam(K);
                nfl_cam_ray(*cam, cam->pix0 + ray, r0, r1);
            } else
            {
                const float* rp = K->a.d_rays + (size_t)ray * 8;
                r0 = *reinterpret_cast<const f4v*>(rp);
                r1 = *reinterpret_cast<const f4v*>(rp + 4);
            }
            const float near = r1[2], far = r1[3];
            const float z = nfl_z_at(K->a, ray, near, far, ii);
            const float zn = ii + 1 < N ? nfl_z_at(K->a, ray, near, far, ii + 1) : z;
            // padded segments recompute (and re-store) the last real one: identical bytes, no branch
            // stash k-step image = [sample c][lane half h][8 values]: a sample's 16 features are 32 contiguous
            // bytes, which makes the weight-gradient kernel's transposed LDS reads conflict-free
            st[cb] = STASH ? K->a.d_act_stash + ((size_t)(ray0 * SPR + gg) * nfl_act_rec(NKP, SMULT)) * 1024 + (2 * c + h) * 16
                           : nullptr;
            mst[cb] = STASH ? K->a.d_act_stash + nfl_msk_offset((size_t)K->a.n_rays * SPR, NKP, SMULT)
                                  + (size_t)(ray0 * SPR + gg) * (NFL_MSK_WORDS * 256) + lane * 16
                            : nullptr;
            s_ray[cb] = ray;
            s_idx[cb] = ii;
            s_ok[cb] = ok;
            s_z[cb] = z;
            s_dl[cb] = ii + 1 < N ? zn - z : 1e2f;
            if (K->a.d_z_out && ok && h == 0) K->a.d_z_out[(size_t)ray * N + ii] = z;
            float raw[3], th[3], tl[3];
            raw[0] = r0[0] + r0[3] * z;
            raw[1] = r0[1] + r1[0] * z;
            raw[2] = r0[2] + r1[1] * z;
#pragma unroll
            for (int k = 0; k < 3; ++k) nfl_turns(raw[k], th[k], tl[k]);
#pragma unroll
            for (int ks = 0; ks < NKP; ++ks) {
                nfl_pe_kstep<NFX, NP, LO>(ks, h, raw, th, tl, pw_lds, P[ks][cb], STASH ? st[cb] + ks * 1024 : nullptr);
                __builtin_amdgcn_sched_barrier(0);      // bound the register pressure of the encoder
            }
        }

        // ------------------------------------------------------------ the field
        K = nfl_kargs();
        int rt = 0;
        // raw head outputs of sample c (lane half 0); extracted at once so the 16-register
        // accumulator tiles die immediately
        float o_sig[NCB], o_rgb[NCB][3], o_tr[NCB][5];
        NFL_STAMP(0);
        nfl_dense<NP, NCB, NKP, 0, true, 8, 2, STASH, LO, NFL_PRODS[NFL_P_IDX_L1]>(ring, bias_lds, rt, h, P, 0, P, 0, X, 0, st, nfl_act_h(NKP, 1), mst, nfl_msk_h(1));       // L1
        NFL_STAMP(1);
        nfl_dense<NP, NCB, 16, 0, true, 8, 1, STASH, LO, NFL_PRODS[NFL_P_IDX_L2]>(ring, bias_lds, rt, h, X, 0, X, 0, Y, 0, st, nfl_act_h(NKP, 2), mst, nfl_msk_h(2));        // L2
        NFL_STAMP(2);
        nfl_dense<NP, NCB, 16, 0, true, 8, 1, STASH, LO, NFL_PRODS[NFL_P_IDX_L3]>(ring, bias_lds, rt, h, Y, 0, Y, 0, X, 0, st, nfl_act_h(NKP, 3), mst, nfl_msk_h(3));        // L3
        NFL_STAMP(3);
        nfl_dense<NP, NCB, 16, 0, true, 8, 1, STASH, LO, NFL_PRODS[NFL_P_IDX_L4]>(ring, bias_lds, rt, h, X, 0, X, 0, Y, 0, st, nfl_act_h(NKP, 4), mst, nfl_msk_h(4));        // L4
        NFL_STAMP(4);
        nfl_dense<NP, NCB, NKP, 16, true, 8, 1, STASH, LO, NFL_PRODS[NFL_P_IDX_L5]>(ring, bias_lds, rt, h, P, 0, Y, 0, X, 0, st, nfl_act_h(NKP, 5), mst, nfl_msk_h(5));      // L5 (skip)
        NFL_STAMP(5);
        nfl_dense<NP, NCB, 16, 0, true, 8, 1, STASH, LO, NFL_PRODS[NFL_P_IDX_L6]>(ring, bias_lds, rt, h, X, 0, X, 0, Y, 0, st, nfl_act_h(NKP, 6), mst, nfl_msk_h(6));        // L6
        NFL_STAMP(6);
        nfl_dense<NP, NCB, 16, 0, true, 8, 1, STASH, LO, NFL_PRODS[NFL_P_IDX_L7]>(ring, bias_lds, rt, h, Y, 0, Y, 0, X, 0, st, nfl_act_h(NKP, 7), mst, nfl_msk_h(7));        // L7
        NFL_STAMP(7);
        nfl_dense<NP, NCB, 16, 0, true, 8, 1, STASH, LO, NFL_PRODS[NFL_P_IDX_L8]>(ring, bias_lds, rt, h, X, 0, X, 0, Y, 0, st, nfl_act_h(NKP, 8), mst, nfl_msk_h(8));        // L8
        NFL_STAMP(8);
        {
            f16v hacc[NCB];
            nfl_head<NP, NCB, 16, NFL_PRODS[NFL_P_IDX_SIG]>(ring, bias_lds, rt, h, Y, 0, hacc);   // sigma
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) o_sig[cb] = hacc[cb][0];
        }
        NFL_STAMP(9);
        K = nfl_kargs();           // head-side arguments (view directions, latents): loaded after the trunk
        if (!K->a.sigma_only) {
            // xyz_encoding_final has no tiles: it is linear and folded into the 256 columns of dir_encoding.0 /
            // transient_encoding.0 that read its output (nfl_plan.cpp), so both read h8 (Y) directly and write into X
            NFL_STAMP(10);
            {
                h8 D[5][NCB][NP];
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) {
                    if constexpr (EMBED) {
                        const int b = s_ray[cb] * 32 + s_idx[cb];
                        const float* xr = K->a.d_embedded + (size_t)(b < K->n_points ? b : K->n_points - 1) * K->emb_stride
                                          + K->cx;
#pragma unroll
                        for (int ks = 0; ks < 5; ++ks) {
                            if (ks >= 2 && !K->has_a) break;
                            float v[8];
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                const int f = 16 * (ks < 2 ? ks : ks - 2) + 8 * h + j;
                                v[j] = ks < 2 ? (f < K->cd ? xr[f] : 0.f) : (f < K->n_a ? xr[K->cd + f] : 0.f);
                            }
                            nfl_split8<NP>(v, D[ks][cb]);
                        }
                        continue;
                    }
                    float raw[3], th[3], tl[3];
                    if (K->gen_rays && !K->a.d_view_dir) {
                        f4v g0, g1;
                        const NflKCam cam = nfl_kcam(K);
                        nfl_cam_ray(*cam, cam->pix0 + s_ray[cb], g0, g1);
                        raw[0] = g0[3];
                        raw[1] = g1[0];
                        raw[2] = g1[1];
                    } else {
                        const float* dp = K->a.d_view_dir ? K->a.d_view_dir + (size_t)s_ray[cb] * 3
                                                       : K->a.d_rays + (size_t)s_ray[cb] * 8 + 3;
#pragma unroll
                        for (int k = 0; k < 3; ++k) raw[k] = dp[k];
                    }
#pragma unroll
                    for (int k = 0; k < 3; ++k) nfl_turns(raw[k], th[k], tl[k]);
                    char* sd = STASH ? st[cb] + nfl_act_d(NKP) * 1024 : nullptr;
                    nfl_pe_kstep<4, NP, LO>(0, h, raw, th, tl, pw_lds + 16, D[0][cb], sd);
                    __builtin_amdgcn_sched_barrier(0);
                    nfl_pe_kstep<4, NP, LO>(1, h, raw, th, tl, pw_lds + 16, D[1][cb], STASH ? sd + 1024 : nullptr);
                    __builtin_amdgcn_sched_barrier(0);
                    if (K->has_a) {
                        const int na = K->n_a;
                        const float* ap = K->a.d_a_emb + (size_t)s_ray[cb] * na + 8 * h;
#pragma unroll
                        for (int ks = 0; ks < 3; ++ks) {
                            float v[8];
                            if (na == 48) {          // the reference's default width: rows are 16-byte aligned
                                const f4v v0 = *reinterpret_cast<const f4v*>(ap + 16 * ks);
                                const f4v v1 = *reinterpret_cast<const f4v*>(ap + 16 * ks + 4);
                                v[0] = v0[0]; v[1] = v0[1]; v[2] = v0[2]; v[3] = v0[3];
                                v[4] = v1[0]; v[5] = v1[1]; v[6] = v1[2]; v[7] = v1[3];
                            } else {                  // narrower code: scalar loads, zeros beyond its width
#pragma unroll
                                for (int j = 0; j < 8; ++j) v[j] = 16 * ks + 8 * h + j < na ? ap[16 * ks + j] : 0.f;
                            }
                            nfl_split8<NP>(v, D[2 + ks][cb]);
                            if (STASH) nfl_stash8<LO>(v, sd + (2 + ks) * 1024);
                        }
                    }
                }
                NFL_STAMP(11);
                if (K->has_a)
                    nfl_dense<NP, NCB, 16, 5, true, 4, 1, STASH, LO, NFL_PRODS[NFL_P_IDX_DIR]>(ring, bias_lds, rt, h, Y, 0, D, 0, X, 0, st, nfl_act_dirh(NKP), mst, nfl_msk_dirh());
                else
                    nfl_dense<NP, NCB, 16, 2, true, 4, 1, STASH, LO, NFL_PRODS[NFL_P_IDX_DIR]>(ring, bias_lds, rt, h, Y, 0, D, 0, X, 0, st, nfl_act_dirh(NKP), mst, nfl_msk_dirh());
            }
            NFL_STAMP(12);
            {
                f16v hacc[NCB];
                nfl_head<NP, NCB, 8, NFL_PRODS[NFL_P_IDX_RGB]>(ring, bias_lds, rt, h, X, 0, hacc);
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) {
                    o_rgb[cb][0] = hacc[cb][0];
                    o_rgb[cb][1] = hacc[cb][1];
                    o_rgb[cb][2] = hacc[cb][2];
                }
            }
            NFL_STAMP(13);
            K = nfl_kargs();
            if (K->use_t) {
                h8 T[1][NCB][NP];
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) {
                    const int tb = s_ray[cb] * 32 + s_idx[cb];
                    const float* tp = EMBED ? K->a.d_embedded + (size_t)(tb < K->n_points ? tb : K->n_points - 1) * K->emb_stride
                                                  + K->cx + K->cd + (K->has_a ? K->n_a : 0) + 8 * h
                                            : K->a.d_t_emb + (size_t)s_ray[cb] * K->n_tau + 8 * h;
                    float v[8];
                    if (EMBED || K->n_tau != 16) {  // rows of the encoded matrix / of a narrower code are not 16-byte aligned
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = 8 * h + j < K->n_tau ? tp[j] : 0.f;
                    } else {
                        const f4v v0 = *reinterpret_cast<const f4v*>(tp);
                        const f4v v1 = *reinterpret_cast<const f4v*>(tp + 4);
                        v[0] = v0[0]; v[1] = v0[1]; v[2] = v0[2]; v[3] = v0[3];
                        v[4] = v1[0]; v[5] = v1[1]; v[6] = v1[2]; v[7] = v1[3];
                    }
                    nfl_split8<NP>(v, T[0][cb]);
                    if (STASH) nfl_stash8<LO>(v, st[cb] + nfl_act_tau(NKP) * 1024);
                }
                nfl_dense<NP, NCB, 16, 1, true, 4, 1, STASH, LO, NFL_PRODS[NFL_P_IDX_T1]>(ring, bias_lds, rt, h, Y, 0, T, 0, X, 8, st, nfl_act_g(NKP, 1), mst, nfl_msk_g(1));
                nfl_dense<NP, NCB, 8, 0, true, 4, 2, STASH, LO, NFL_PRODS[NFL_P_IDX_T2]>(ring, bias_lds, rt, h, X, 8, X, 8, X, 0, st, nfl_act_g(NKP, 2), mst, nfl_msk_g(2));
                nfl_dense<NP, NCB, 8, 0, true, 4, 2, STASH, LO, NFL_PRODS[NFL_P_IDX_T3]>(ring, bias_lds, rt, h, X, 0, X, 0, X, 8, st, nfl_act_g(NKP, 3), mst, nfl_msk_g(3));
                nfl_dense<NP, NCB, 8, 0, true, 4, 2, STASH, LO, NFL_PRODS[NFL_P_IDX_T4]>(ring, bias_lds, rt, h, X, 8, X, 8, X, 0, st, nfl_act_g(NKP, 4), mst, nfl_msk_g(4));
                f16v hacc[NCB];
                nfl_head<NP, NCB, 8, NFL_PRODS[NFL_P_IDX_THEAD]>(ring, bias_lds, rt, h, X, 0, hacc);
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
                    for (int k = 0; k < 5; ++k) o_tr[cb][k] = hacc[cb][k];     // row 8 (beta) = register 4
                }
            }
        }

        NFL_STAMP(14);
        // ------------------------------------------------------------ compositing, phase 1
        K = nfl_kargs();
        // (reference models/rendering.py:141-226).  Lanes 0..31 of each half own sample c.
        float w_loc[NCB], sig_t[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const bool ok = s_ok[cb];
            const float z = s_z[cb], dl = s_dl[cb];
            const size_t sidx = (size_t)s_ray[cb] * N + s_idx[cb];
            const float sg = nfl_softplus(o_sig[cb]);
            float cr = 0.f, cg = 0.f, cbl = 0.f, tr = 0.f, tg = 0.f, tb = 0.f, sgt = 0.f, bt = 0.f;
            if (!K->a.sigma_only) {
                cr = nfl_sigmoid(o_rgb[cb][0]);
                cg = nfl_sigmoid(o_rgb[cb][1]);
                cbl = nfl_sigmoid(o_rgb[cb][2]);
            }
            if (K->use_t) {
                sgt = nfl_softplus(o_tr[cb][0]);
                tr = nfl_sigmoid(o_tr[cb][1]);
                tg = nfl_sigmoid(o_tr[cb][2]);
                tb = nfl_sigmoid(o_tr[cb][3]);
                bt = nfl_softplus(o_tr[cb][4]);
            }
            if (K->a.d_field_raw && ok && h == 0) {
                float* fr = K->a.d_field_raw + sidx * 9;
                fr[0] = cr; fr[1] = cg; fr[2] = cbl; fr[3] = sg;
                fr[4] = tr; fr[5] = tg; fr[6] = tb; fr[7] = sgt; fr[8] = bt;
            }
            if constexpr (EMBED) continue;          // the field outputs are the result; nothing to composite
            float alpha, a_s = 0.f, a_t = 0.f;
            if (K->use_t) {
                a_s = 1.f - expf(-dl * sg);
                a_t = 1.f - expf(-dl * sgt);
                alpha = 1.f - expf(-dl * (sg + sgt));
            } else {
                const float nz = K->a.d_noise ? K->a.d_noise[sidx] * K->a.noise_std : 0.f;
                alpha = 1.f - expf(-dl * fmaxf(sg + nz, 0.f));
            }
            if (!ok) { alpha = 0.f; a_s = 0.f; a_t = 0.f; }
            // exclusive product scans of (1 - alpha) over the 32 samples of the segment
            float inc[3] = {1.f - alpha, 1.f - a_s, 1.f - a_t};
#pragma unroll
            for (int d = 1; d < 32; d <<= 1)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float n = __shfl_up(inc[k], d, 32);
                    if (c >= d) inc[k] *= n;
                }
            float exc[3], prod[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float e = __shfl_up(inc[k], 1, 32);
                exc[k] = c == 0 ? 1.f : e;
                prod[k] = __shfl(inc[k], 31, 32);
            }
            const float w = alpha * exc[0];
            w_loc[cb] = w;
            sig_t[cb] = sgt;
            float rec[NFL_NST];
            rec[0] = prod[0]; rec[1] = prod[1]; rec[2] = prod[2];
            rec[3] = nfl_sum32(w);
            rec[11] = nfl_sum32(w * z);
            if (K->use_t) {
                const float ws = a_s * exc[0], wt = a_t * exc[0];
                rec[4] = nfl_sum32(ws * cr); rec[5] = nfl_sum32(ws * cg); rec[6] = nfl_sum32(ws * cbl);
                rec[7] = nfl_sum32(wt * tr); rec[8] = nfl_sum32(wt * tg); rec[9] = nfl_sum32(wt * tb);
                rec[10] = nfl_sum32(wt * bt);
                if (K->a.test_extras) {
                    const float ws1 = a_s * exc[1], wt1 = a_t * exc[2];
                    rec[12] = nfl_sum32(ws1 * cr); rec[13] = nfl_sum32(ws1 * cg); rec[14] = nfl_sum32(ws1 * cbl);
                    rec[15] = nfl_sum32(ws1 * z);
                    rec[16] = nfl_sum32(wt1 * tr); rec[17] = nfl_sum32(wt1 * tg); rec[18] = nfl_sum32(wt1 * tb);
                    rec[19] = nfl_sum32(wt1 * z);
                } else {
#pragma unroll
                    for (int k = 12; k < NFL_NST; ++k) rec[k] = 0.f;
                    if (NFL_LOSS_ON(K->a)) rec[19] = nfl_sum32(ok ? sgt : 0.f);    // s_l: plain sum of the transient densities
                }
            } else {
                rec[4] = nfl_sum32(w * cr); rec[5] = nfl_sum32(w * cg); rec[6] = nfl_sum32(w * cbl);
#pragma unroll
                for (int k = 7; k < NFL_NST; ++k) if (k != 11) rec[k] = 0.f;
            }
            if (lane == 0) {
                float* dst = rec_lds + (wave * NCB + cb) * NFL_REC;
#pragma unroll
                for (int k = 0; k < NFL_NST; ++k) dst[k] = rec[k];
            }
        }
        if constexpr (EMBED) continue;
        NFL_STAMP(15);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        NFL_STAMP(16);

        // ------------------------------------------------------------ compositing, phase 2
        K = nfl_kargs();
        // fold the segments of each ray in order; lane k (< NFL_NST) carries record entry k
        const float* carry_in = rec_lds + (NSLOT + (tile & 1)) * NFL_REC;
        float* carry_out = rec_lds + (NSLOT + ((tile + 1) & 1)) * NFL_REC;
        const int kk = lane < NFL_NST ? lane : 0;
        const int chain = kk < 3 ? kk : (kk < 12 ? 0 : (kk < 16 ? 1 : 2));
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const int m = wave * NCB + cb;
            const int g = tile * NSLOT + m;
            if (g >= seg_end) continue;                         // wave-uniform
            const int q = g % SPR, ray = ray0 + g / SPR;
            int m0 = m - q;
            float t_in = 1.f;        // transmittance (chain 0) entering my segment: every lane
            float t_run = 1.f;       // running product of my chain
            float acc = 0.f;
            if (m0 < 0) {
                t_in = carry_in[0];
                t_run = carry_in[chain];
                acc = kk < 3 ? 0.f : carry_in[kk];
                m0 = 0;
            }
            const bool plain = NFL_LOSS_ON(K->a) && kk == 19;      // entry 19 with the fused loss: an unweighted sum (s_l)
            for (int mm = m0; mm <= m; ++mm) {
                const float* r = rec_lds + mm * NFL_REC;
                if (mm < m) t_in *= r[0];
                acc += (plain ? 1.f : t_run) * r[kk];
                t_run *= r[chain];
            }
            // per-sample outputs
            if (s_ok[cb] && h == 0) {
                const size_t sidx = (size_t)ray * N + s_idx[cb];
                if (K->a.d_weights) K->a.d_weights[sidx] = w_loc[cb] * t_in;
                if (K->use_t && K->a.d_transient_sigmas) K->a.d_transient_sigmas[sidx] = sig_t[cb];
            }
            if (q == SPR - 1) {
                // ray complete: lane k holds the composited quantity k
                const float wsum = __shfl(acc, 3);
                if (K->a.d_status) {       // fp16 operand range exceeded somewhere on this ray (header: d_status)
                    const bool bad = lane < NFL_NST && !(fabsf(acc) <= 3.0e38f);
                    if (__builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) atomicOr(K->a.d_status, NFL_STATUS_NONFINITE);
                }
                const float white = K->a.white_back ? 1.f - wsum : 0.f;
                const float stat = acc + white;                  // meaningful on lanes 4..6 and 12..14
                const float tran = __shfl(acc, (lane + 3) & 63); // lanes 4..6 read 7..9
                if (lane == 3 && K->a.d_opacity) K->a.d_opacity[ray] = acc;
                if (lane == 11 && K->a.d_depth) K->a.d_depth[ray] = acc;
                if (lane >= 4 && lane < 7) {
                    if (K->use_t) {
                        if (K->a.d_rgb_static) K->a.d_rgb_static[ray * 3 + lane - 4] = stat;
                        if (K->a.d_rgb_transient) K->a.d_rgb_transient[ray * 3 + lane - 4] = tran;
                        if (K->a.d_rgb) K->a.d_rgb[ray * 3 + lane - 4] = stat + tran;
                    } else if (K->a.d_rgb) {
                        K->a.d_rgb[ray * 3 + lane - 4] = stat;
                    }
                }
                if (NFL_LOSS_ON(K->a)) {
                    // NerfWLoss of this ray (reference losses.py:35-50) and its backward seeds
                    const bool ch = lane >= 4 && lane < 7;
                    const float tgt = ch ? K->a.d_loss_target[ray * 3 + lane - 4] : 0.f;
                    const float diff = ch ? (K->use_t ? stat + tran : stat) - tgt : 0.f;
                    const float sq = diff * diff;
                    const float sum3 = __shfl(sq, 4) + __shfl(sq, 5) + __shfl(sq, 6);
                    const float rinv = 1.f / (float)K->a.n_rays, c0 = K->a.loss_coef;
                    float l0, l1 = 0.f, l2 = 0.f, g_rgb;
                    if (K->use_t) {
                        const float beta = __shfl(acc, 10) + K->beta_min;
                        const float ib2 = 1.f / (beta * beta);
                        l0 = c0 * sum3 * 0.5f * ib2 * rinv * (1.f / 3.f);
                        l1 = c0 * (3.f + logf(beta)) * rinv;
                        l2 = c0 * K->a.lambda_u * __shfl(acc, 19) * rinv / (float)N;
                        g_rgb = c0 * diff * ib2 * rinv * (1.f / 3.f);
                        if (lane == 10 && K->a.d_seed_beta)
                            K->a.d_seed_beta[ray] = c0 * rinv * (1.f / beta - sum3 * ib2 / beta * (1.f / 3.f));
                    } else {
                        l0 = c0 * 0.5f * sum3 * rinv * (1.f / 3.f);
                        g_rgb = c0 * diff * rinv * (1.f / 3.f);
                    }
                    if (ch && K->a.d_seed_rgb) K->a.d_seed_rgb[ray * 3 + lane - 4] = g_rgb;
                    if (lane == 0) {
                        atomicAdd(loss_lds + K->a.loss_slot, l0);
                        if (K->use_t) {
                            atomicAdd(loss_lds + 2, l1);
                            atomicAdd(loss_lds + 3, l2);
                        }
                    }
                }
                if (K->use_t) {
                    if (lane == 10 && K->a.d_beta) K->a.d_beta[ray] = acc + K->beta_min;
                    if (K->a.test_extras) {
                        if (lane >= 12 && lane < 15 && K->a.d_rgb_static_only) K->a.d_rgb_static_only[ray * 3 + lane - 12] = stat;
                        if (lane == 15 && K->a.d_depth_static_only) K->a.d_depth_static_only[ray] = acc;
                        if (lane >= 16 && lane < 19 && K->a.d_rgb_transient_only) K->a.d_rgb_transient_only[ray * 3 + lane - 16] = acc;
                        if (lane == 19 && K->a.d_depth_transient_only) K->a.d_depth_transient_only[ray] = acc;
                    }
                }
            } else if (m == NSLOT - 1) {
                // the ray continues in the next tile: hand its state over
                if (lane < NFL_NST) carry_out[lane] = lane < 3 ? t_run : acc;
            }
        }
        // the next tile's first consume() barrier orders these LDS reads/writes
        // against the next phase-1 record writes (>= 70 barriers away).
        NFL_STAMP(17);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the two prefetched chunks before exit
    NflKArgs K = nfl_kargs();
    if (NFL_LOSS_ON(K->a)) {   // one flush of this workgroup's loss partials
        __syncthreads();
        if (tid < 4 && K->a.d_losses && loss_lds[tid] != 0.f) atomicAdd(K->a.d_losses + tid, loss_lds[tid]);
    }
    if (K->a.d_status) {      // an activation beyond fp16's range: the conversion gave inf (0x7c00) -- header, d_status
        const bool bad = (ring.ovf & 0xffffu) >= 0x7c00u || (ring.ovf >> 16) >= 0x7c00u;
        if (__builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) atomicOr(K->a.d_status, NFL_STATUS_RANGE);
    }
#ifdef NFL_STAMPS
    NFL_STAMP(18);
    t_acc[14] = ring.t_wait;
    t_acc[19] = ring.t_bar;
    if (lane == 0 && blockIdx.x < 1024)
        for (int i = 0; i < NFL_NSTAMP; ++i) nfl_stamp_buf[(blockIdx.x * 4 + wave) * NFL_NSTAMP + i] = t_acc[i];
#endif
}

template <int NSPLIT, int NCB, int NFX, int MODE>
static int nfl_launch_render_t(const NflPlan* hp, const void* d_plan, const void* d_packed,
                             const nfl_pass_args* args, hipStream_t stream) {
    using C = NflRenderCfg<NSPLIT, NCB, NFX>;
    RenderArgs A;
    A.plan = static_cast<const NflPlan*>(d_plan);
    A.packed = static_cast<const char*>(d_packed);
    A.a = *args;
    A.has_a = hp->has_a;
    A.use_t = (hp->has_t && args->d_t_emb != nullptr && !args->sigma_only) ? 1 : 0;
    A.n_points = 0;
    A.emb_stride = 0;
    A.gen_rays = 0;
    A.nfx_rt = hp->n_emb_xyz;
    A.ndir_rt = (hp->ld[NFL_P_DIR] - NFL_W - hp->n_a - 3) / 6;
    A.cx = 6 * A.nfx_rt + 3;
    A.cd = 6 * A.ndir_rt + 3;
    A.n_a = hp->n_a;
    A.n_tau = hp->n_tau;
    memset(&A.cam, 0, sizeof(A.cam));
    if ((args->h_cam != nullptr || args->d_cam != nullptr) && MODE != NFL_MODE_EMBED) {
        A.gen_rays = 1;
        if (args->h_cam != nullptr && args->d_cam == nullptr) A.cam = *args->h_cam;
    }
    A.a.h_cam = nullptr;                  // a host pointer has no business on the device
    if (MODE == NFL_MODE_EMBED) {      // n_rays = segments of 32 points; d_t_emb != NULL only flags "transient head on"
        A.n_points = args->n_points;
        A.emb_stride = args->embedded_stride;
    }
    A.n_chunks = args->sigma_only ? hp->n_chunks_sigma : (A.use_t ? hp->n_chunks : hp->n_chunks_static);
    A.n_rt = args->sigma_only ? hp->n_rt_sigma : (A.use_t ? hp->n_rt : hp->n_rt_static);
    A.bias_off = hp->bias_off;
    A.spr = (args->n_samples + 31) / 32;
    A.beta_min = hp->beta_min;
    // contiguous ray ranges, one workgroup per CU where there is enough work
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    }
    int rpw = (args->n_rays + ncu - 1) / ncu;
    // keep whole tiles per workgroup where possible: rays per tile = NSLOT / spr (if >= 1)
    const int rays_per_tile = C::NSLOT / A.spr;
    if (rays_per_tile > 1) rpw = (rpw + rays_per_tile - 1) / rays_per_tile * rays_per_tile;
    if (rpw < 1) rpw = 1;
    A.rays_per_wg = rpw;
    const int grid = (args->n_rays + rpw - 1) / rpw;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&nfl_render_kernel<NSPLIT, NCB, NFX, MODE>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES) != hipSuccess)
            return NFL_ENODEV;
        attr_set = true;
    }
    hipLaunchKernelGGL((nfl_render_kernel<NSPLIT, NCB, NFX, MODE>), dim3(grid), dim3(256), C::LDS_BYTES, stream, A);
    return hipGetLastError() == hipSuccess ? NFL_OK : NFL_ELAUNCH;
}

template <int NSPLIT, int NCB, int NFX>
static int nfl_launch_render(const NflPlan* hp, const void* d_plan, const void* d_packed,
                             const nfl_pass_args* args, hipStream_t stream) {
#ifdef NFL_DIAG_INFERENCE_ONLY      // sweep builds (nfl_diag.h): only the inference instantiation is compiled
    if (args->d_embedded || args->d_act_stash) return NFL_EINVAL;
    return nfl_launch_render_t<NSPLIT, NCB, NFX, NFL_MODE_RENDER>(hp, d_plan, d_packed, args, stream);
#endif
    if (args->d_embedded) return nfl_launch_render_t<NSPLIT, NCB, NFX, NFL_MODE_EMBED>(hp, d_plan, d_packed, args, stream);
    if (args->d_act_stash) {
        if (NSPLIT != 3) return NFL_EINVAL;        // the training stash is written by the accurate mode only
        if (args->stash_split)
            return nfl_launch_render_t<NSPLIT, NCB, NFX, (NSPLIT == 3 ? NFL_MODE_STASH2 : NFL_MODE_RENDER)>(hp, d_plan, d_packed, args, stream);
        return nfl_launch_render_t<NSPLIT, NCB, NFX, (NSPLIT == 3 ? NFL_MODE_STASH : NFL_MODE_RENDER)>(hp, d_plan, d_packed, args, stream);
    }
    return nfl_launch_render_t<NSPLIT, NCB, NFX, NFL_MODE_RENDER>(hp, d_plan, d_packed, args, stream);
}
