// nfl_plan.h -- the static schedule of one field's packed weight stream.
//
// The fused render kernel evaluates the MLP transposed, H^T[out, sample] =
// W[out, in] * H^T[in, sample], one 32-row "row tile" of W at a time, with
// v_mfma_f32_32x32x16_f16.  W is the A operand; it is pre-packed on the device
// into MFMA fragment order ("frags": 64 lanes x 8 halves = 1 KiB; lane l holds
// row l&31, k-slots 8*(l>>5)+j) in exactly the order the kernel consumes it, so
// the kernel streams it global -> LDS -> registers with linear, conflict-free
// accesses.  This header is shared by the host plan builder, the pack kernel
// and the render kernel; the three must agree on the order below.
//
// Stream order (row tiles; "ks" = k-step of 16 input columns):
//   L1      8 row tiles, K = P (encoded xyz, NKP ks)              -> X, relu
//   L2..L4  8 row tiles each, K = 16 ks                           X->Y->X->Y
//   L5      8 row tiles, K = P + Y (skip: encoded xyz FIRST)      -> X, relu
//   L6..L8  8 row tiles each                                      X->Y->X->Y
//   SIG     1 row tile (row 0 = static_sigma), K = Y              -> sigma
//   ---- a sigma-only pass stops here ----
//   (xyz_encoding_final has no tiles: linear, folded into the 256 columns of DIR / T1 that read its output)
//   DIR     4 row tiles, K = Y + D (dir PE 2 ks [+ appearance 3 ks]) -> X[0:8], relu
//   RGB     1 row tile (rows 0..2), K = X[0:8]                    -> rgb
//   ---- a pass without the transient head stops here ----
//   T1      4 row tiles, K = Y + tau (1 ks)                       -> X[8:16], relu
//   T2..T4  4 row tiles each, K = 8 ks               X[8:16]->X[0:8]->X[8:16]->X[0:8]
//   THEAD   1 row tile (row 0 sigma_t, 1..3 rgb_t, 8 beta), K = X[0:8]
//
// Chunks (the unit of the LDS ring; every chunk starts a new barrier epoch):
//   L1: 2 row tiles per chunk; T2..T4: 2 row tiles per chunk; otherwise 1.
#pragma once
#include <stddef.h>
#include <stdint.h>

#define NFL_PLAN_MAGIC 0x4e464c31u /* "NFL1" */
#if defined(__HIPCC__)
#define NFL_HD_EARLY __host__ __device__ inline constexpr
#else
#define NFL_HD_EARLY inline constexpr
#endif
#define NFL_W 256
#define NFL_MAX_RT 112
#define NFL_MAX_CHUNKS 104

// k-slot -> input-column maps
#define NFL_SEG_ACT 0   // produced by a previous row tile: k-slot (ks,h,j) <-> column 32*(ks>>1) + 16*(ks&1) + 8*(j>>2) + 4*h + (j&3)
#define NFL_SEG_NAT 1   // built directly by the lanes:      k-slot (ks,h,j) <-> column 16*ks + 8*h + j

// Encoder widths (reference opt.py:25-28: --N_emb_xyz / --N_emb_dir are free integers).  The kernels are instantiated
// for two position-encoding k-step counts: 4 (up to 10 frequencies, 63 features) and 6 (up to 15, 93).  A narrower
// encoder runs in the next wider instantiation: the kernel still evaluates all its sin / cos features, but the packed
// weights of the columns beyond 6 N + 3 are zero (the packer pads), so they contribute nothing -- forward, dgrad and wgrad
// alike (their tile descriptors carry the true column counts).  The direction encoder has two k-steps: N_emb_dir <= 4.
#define NFL_MAX_EMB_XYZ 15
#define NFL_MAX_EMB_DIR 4
NFL_HD_EARLY int nfl_kernel_nfx(int n_emb_xyz) { return n_emb_xyz <= 10 ? 10 : 15; }       // instantiation that runs it
NFL_HD_EARLY int nfl_nkp_for(int n_emb_xyz) { return (6 * nfl_kernel_nfx(n_emb_xyz) + 3 + 15) / 16; }

struct NflBlk {      // rows [src_row0, src_row0+nrows) of layer `layer` land on tile rows dst_row..
    int16_t layer, nrows, src_row0, dst_row;
};
struct NflSeg {      // `nks` k-steps reading columns col0 + map(k-slot), valid while map < ncols.
                     // Transposed tiles (dgrad streams): the k-slots run over ROWS col0 + map of `layer`.
    int16_t nks, kind, col0, ncols;
    int16_t layer, pad0, pad1, pad2;
};
struct NflRowTile {
    int32_t frag_off;   // index of this tile's first k-step in the stream (units of k-steps)
    int32_t nks;        // total k-steps
    int32_t nblk, nseg;
    // forward tiles (trans == 0): element(i, m) = W[blk(i).layer][blk.src_row0 + i - blk.dst_row][seg.col0 + m]
    // dgrad tiles   (trans == 1): element(i, m) = W[seg.layer][seg.col0 + m][tcol0 + i]  (i < tncols), bias 0
    int16_t trans, tcol0, tncols, pad;
    NflBlk blk[3];
    NflSeg seg[3];
};

struct NflPlan {
    uint32_t magic;
    int32_t prec, nsplit;         // nsplit = 1 or 3 products; frags carry (nsplit==3 ? hi+lo : hi)
    int32_t elem;                 // 0: fp16 fragments (both streams now), 1: bf16 fragments
    int32_t is_bwd;               // 1: this is the dgrad (transposed) stream
    int32_t reserved_flags;       // dgrad stream: bit 0 = carries the tiles for the gradient w.r.t. the rays; bits 8..15: n_emb_dir
    int32_t n_emb_xyz, nkp;       // nkp = k-steps of the encoded position in the kernel instantiation that runs this field (nfl_nkp_for)
    int32_t has_a, has_t, n_a, n_tau;
    int32_t n_rt, n_rt_sigma, n_rt_static;
    int32_t n_chunks, n_chunks_sigma, n_chunks_static;   // dgrad stream: n_chunks_sigma = first chunk after the transient head's
    int32_t total_ks;             // k-steps in the whole stream
    int32_t ks_bytes;             // bytes per k-step: 1024 (hi) or 2048 (hi+lo)
    int32_t max_chunk_ks;         // largest chunk, in k-steps
    int32_t stream_bytes;         // total_ks * ks_bytes
    int32_t bias_off;             // byte offset of the fp32 bias table [n_rt][32] inside the packed buffer
    int32_t packed_bytes;
    float   beta_min;
    int32_t ld[19];               // in_features of each layer (row stride of its weight)
    int32_t chunk_off[NFL_MAX_CHUNKS + 1];   // byte offset of each chunk in the stream (+ end)
    int32_t chunk_aux[NFL_MAX_CHUNKS + 1];   // dgrad: relu-mask word (nfl_msk_*) of the chunk's first tile; -1 = none
    NflRowTile rt[NFL_MAX_RT];
};

#if defined(__HIPCC__)
#define NFL_HD __host__ __device__ inline
#else
#define NFL_HD inline
#endif
// ---- per-segment (32 samples) stash records, in k-steps of 1 KiB (64 lanes x 8 fp16) ----
// forward activations (inputs of every layer), written by the training-mode forward:
//   P | h1..h8 | D (dir PE 2, appearance 3) | dirh | tau | g1..g4
// (no `feat`: xyz_encoding_final is linear and the gradients that would read its output are composed from
//  sum_s delta_dirh (x) h8 instead, nfl_wgrad.hip)
NFL_HD constexpr int nfl_act_h(int nkp, int l) { return nkp + 16 * (l - 1); }   // l = 1..8
NFL_HD constexpr int nfl_act_d(int nkp) { return nkp + 128; }
NFL_HD constexpr int nfl_act_dirh(int nkp) { return nkp + 133; }
NFL_HD constexpr int nfl_act_tau(int nkp) { return nkp + 141; }
NFL_HD constexpr int nfl_act_g(int nkp, int m) { return nkp + 142 + 8 * (m - 1); }   // m = 1..4
NFL_HD constexpr int nfl_act_slots(int nkp) { return nkp + 174; }
// Split (hi + lo) stashes of the fp32-class backward (NFL_BWD_F16X3): every record is followed by a second one of the
// same layout holding the fp16 residuals, i.e. a segment's record is `mult` = 2 times as long and the residual of the
// k-step in slot s sits nfl_act_slots(nkp) (NFL_GRD_SLOTS) slots behind it -- a compile-time offset for the writers,
// a second base pointer for the weight-gradient GEMMs.
NFL_HD constexpr int nfl_act_rec(int nkp, int mult) { return nfl_act_slots(nkp) * mult; }
// relu masks, written by the training-mode forward behind the activation records (one 32-bit word per lane and
// row tile of a relu layer: bit 2p = value 2p of the lane's 16 accumulators was positive, bit 16+2p = value 2p+1;
// 256 B per wave and tile instead of the 2 KiB of fp16 activations the dgrad kernel would otherwise re-read;
// record layout [group of 4 tiles][lane][4 words], so one 16 B store / one 1 KiB DMA piece moves four tiles' words):
//   h1..h8 (8 tiles each) | dirh (4) | g1..g4 (4 each)
NFL_HD constexpr int nfl_msk_h(int l) { return 8 * (l - 1); }        // l = 1..8
NFL_HD constexpr int nfl_msk_dirh() { return 64; }
NFL_HD constexpr int nfl_msk_g(int m) { return 68 + 4 * (m - 1); }   // m = 1..4
#define NFL_MSK_WORDS 84
// byte offset of the mask records inside the activation stash buffer (after the records and their 4 KiB tail pad)
NFL_HD constexpr size_t nfl_msk_offset(size_t n_seg, int nkp, int mult = 1) { return n_seg * (size_t)nfl_act_rec(nkp, mult) * 1024 + 4096; }
// pre-activation gradients, written by the dgrad kernel (d(feat) lives in registers only, see above):
//   d1..d8 | ddirh | dg1..dg4 | head grads as natural k-steps: dsigma, drgb, dsigma_t, drgb_t, dbeta
#define NFL_GRD_D(l) (16 * ((l) - 1))
#define NFL_GRD_DIRH 128
#define NFL_GRD_G(m) (136 + 8 * ((m) - 1))
#define NFL_GRD_HEADS 168
#define NFL_GRD_SLOTS 173

#define NFL_GMAX_SLOTS 1024     // d_gmax is 1024 floats: workgroups spread their atomicMax over them, readers take the max
// Loss scale of a backward pass: the power of two that brings max|head gradient| (bits of the fp32 the
// compositing backward left in d_gmax) to [2^5, 2^6).  fp16 overflows at 65504, so intermediate gradients may
// grow 1024x over the largest head gradient; anything 2^-19 below it is still a normal fp16 (2^-29: subnormal).
// dgrad and wgrad both derive the scale from the same words, so they agree bit for bit.
#ifndef NFL_LOSS_SCALE_LOG2
#define NFL_LOSS_SCALE_LOG2 5
#endif
NFL_HD float nfl_loss_scale_from_bits(unsigned bits) {
    int e = (int)((bits >> 23) & 0xffu) - 127;        // floor(log2(gmax)) for normal values
    if (bits == 0u || e < -100) return 1.0f;           // no gradient at all (or denormal): nothing to scale
    if (e > 100) e = 100;
    const unsigned sbits = (unsigned)(127 + NFL_LOSS_SCALE_LOG2 - e) << 23;
    union { unsigned u; float f; } cvt;
    cvt.u = sbits;
    return cvt.f;
}

#ifdef __cplusplus
extern "C" {
#endif
struct nfl_field_desc;
// returns 0 or a negative NFL_E* code
int nfl_plan_fill(const struct nfl_field_desc* desc, int prec, struct NflPlan* plan);
int nfl_plan_fill_bwd(const struct nfl_field_desc* desc, int rays_grad, int bwd_prec, struct NflPlan* plan);
#ifdef __cplusplus
}
#endif
