// nfl_dgrad.hip -- fused backward of the field MLP w.r.t. its activations (dgrad).
//
// Hand-written replacement of what autograd replays for reference models/nerf.py:
// 153-212: for every sample, walk the network from the heads back to layer 2,
//   delta_{l-1} = (W_l^T delta_l) (.) [h_{l-1} > 0]
// with the same register-resident transposed formulation as the forward kernel
// (nfl_render_impl.h): delta^T[feature, sample] tiles are MFMA accumulators, converted
// to fp16 they are the B operand of the next product; W^T streams through the LDS ring
// as pre-packed fp16 fragments.  Mixed precision with a LOSS SCALE: every gradient in
// this kernel is multiplied by the power of two S = nfl_loss_scale_from_bits(*d_gmax)
// (max |head gradient| -> [2^5, 2^6)), so that fp16's 5-bit exponent is spent around the
// values that matter whatever the magnitude of the loss; everything that leaves the
// kernel in fp32 (latent / ray gradients) is multiplied by 1/S, and the gradient stash
// keeps the scaled fp16 values for the weight-gradient GEMMs, which divide by S at the end.
// The relu masks come from the forward pass' mask records (one 32-bit word per lane and row
// tile, nfl_plan.h): the 256 B a wave needs per row tile and segment are DMA'd into the ring
// slot beside the weights, so the kernel issues no ordinary global load inside the layer loop.  Every delta_l is written (fp16, scaled, fragment
// order) to the gradient stash for the weight-gradient GEMMs (nfl_wgrad.hip).  The
// appearance / transient latent gradients are the extra rows of W_dir^T / W_t0^T,
// reduced over the samples of the ray with wave shuffles and accumulated with fp32 atomics.
#include "nfl_render_impl.h"

// Three arithmetics, one kernel template (M = DgMode<NP, NWP>: parts of a gradient operand, fragments of a weight k-step):
//   <1, 1>  NFL_PREC_F16 (default): W_hi d_hi, one product.  Two 32-sample segments (column blocks) per wave and two row
//           tiles per ring chunk: a row tile is only 16 MFMAs per column block, so the per-tile fixed costs (barrier, weight
//           DMA, LDS reads of the A fragments) are shared; the register file holds it because the walk needs only TWO
//           16-k-step operand sets (P, Q below).  Fastest.  Rounded to nearest, W_hi is the same wrong matrix for hundreds of
//           steps in a row once the learning rate has decayed -- a fixed-pattern perturbation of the backward operator that
//           Adam integrates into a systematic offset of long training curves (profiles/r03_psnr_backward_attribution.txt) --
//           so nfl_pack.hip DRAWS the rounding of this stream (stochastic, anew whenever a weight moves), as the epilogue
//           below draws the gradients' (dg_sr_pack): most of the offset goes; a rest stays on the NeRF-W scene.
//   <1, 2>  NFL_PREC_F16W: W_hi d_hi + W_lo d_hi -- the chain sees the weights to fp32 class, the gradients stay single
//           fp16 images (stashes as above), rounded stochastically.  2 KiB per k-step, so one row tile per chunk.  No offset left.
//   <2, 2>  NFL_PREC_F16X3: gradients split hi + lo as well, three products (the forward's f16x3 arithmetic), split
//           gradient stash; the operand sets are twice as large, so one segment per wave.
template <int NP_, int NWP_>
struct DgMode {
    static constexpr int NP = NP_, NWP = NWP_;
    static constexpr int NCB = NP_ == 1 ? 2 : 1;
    static constexpr int TPC = NWP_ == 1 ? 2 : 1;
#ifdef NFL_DIAG_X3_PRODS
    static constexpr int PRODS = NP_ == 2 ? NFL_DIAG_X3_PRODS : (NWP_ == 2 ? 5 : 0);
#else
    static constexpr int PRODS = NP_ == 2 ? 3 : (NWP_ == 2 ? 5 : 0);     // nfl_tile_p: bit 2 = hi + lo weight fragments under single-image operands
#endif
};

struct DgradArgs {
    const NflPlan* plan;
    const char* packed;
    nfl_dgrad_args a;
    int n_chunks, c_start;
    int has_a, has_t, use_t;
    int spr, rays_per_wg, nkp, n_seg_total;
    int rays_tiles;           // the stream carries the encoded-position / direction rows (gradient w.r.t. rays)
    int nfx_rt, ndir_rt;      // the field's frequency counts (<= the instantiation's; nfl_plan.h, "Encoder widths")
    int n_a, n_tau;           // widths of the latent codes (<= 48 / 16)
};

struct DgradArgs;
typedef const __attribute__((address_space(4))) DgradArgs* NflDgKArgs;
NFL_DEV NflDgKArgs nfl_dg_kargs() {
    NflDgKArgs p = (NflDgKArgs)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}

template <int NFX, class M>
struct NflDgradCfg {
    static constexpr int NP = M::NP, NCB = M::NCB, TPC = M::TPC;
    static constexpr int NKP = (6 * NFX + 3 + 15) / 16;
    static constexpr int KSB = 1024 * M::NWP;            // bytes of a weight k-step: hi (+ lo)
    static constexpr int MAXKS = TPC * 17;               // row tiles per chunk
    static constexpr int WBYTES = MAXKS * KSB;
    static constexpr int AUXB = 4 * NCB * 1024;
    static constexpr int SLOT = WBYTES + AUXB;
    static constexpr int MAXP = (WBYTES + 4095) / 4096;
    static constexpr int LDS_TAB = (2 * (NFL_MAX_CHUNKS + 8) + 32) * 4;
    static constexpr int LDS_BYTES = LDS_TAB + 3 * SLOT;
};

// ring with the per-wave mask pieces: pieces 0..MAXPW-1 are weights, the next one per column block the 1 KiB
// (four dwords per lane) of that segment's relu-mask words for a group of four tiles, issued with the group's
// first tile only
template <int SLOT_BYTES, int WBYTES, int MAXPW, int NCB>
struct NflRingAux {
    static constexpr int MAXP = MAXPW + NCB;
    const char* gsrc;
    const int* chunk_off;
    const int* chunk_aux;
    char* lds;
    const char* aux_src;      // mask records (wave-uniform base)
    size_t seg_stride;        // bytes per segment record
    int n_chunks, c_start, c_issue, s_issue, s_read;
    int seg_issue, seg_last;  // this wave's first (clamped) global segment for the tile c_issue belongs to
    int wave, lane;
    const char* i_src;
    const char* i_aux[NCB];
    bool i_mask;
    char* i_dst;
    int i_nbytes;

    int n_off0, n_off1, n_aux;    // table entries of chunk c_issue, fetched one step ahead
    // software count of this wave's VMEM operations, and its value right after the last piece of each chunk in
    // flight: consume() may leave everything younger than that piece outstanding (pieces of the next chunk AND
    // the stash stores issued since -- an HBM write acknowledgement takes longer than a 16-MFMA row tile).
    // Operations that are not counted (loads, atomics, ...) only make the wait stricter.
    int ops, mk0, mk1;        // mk0 / mk1: the count after the last piece of the oldest / youngest chunk in flight
    NFL_DEV void note(int n) { ops += n; }

    NFL_DEV void fetch_tables() {
        n_off0 = __builtin_amdgcn_readfirstlane(chunk_off[c_issue]);
        n_off1 = __builtin_amdgcn_readfirstlane(chunk_off[c_issue + 1]);
        n_aux = __builtin_amdgcn_readfirstlane(chunk_aux[c_issue]);
    }
    NFL_DEV void begin_issue() {
        i_nbytes = n_off1 - n_off0;
        i_src = gsrc + n_off0;
        i_dst = lds + s_issue * SLOT_BYTES;
        const int slot = n_aux < 0 ? 0 : n_aux;
        i_mask = n_aux >= 0 && (n_aux & 3) == 0;          // this chunk opens a group of four masked tiles
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const int sg = seg_issue + cb < seg_last ? seg_issue + cb : seg_last;
            i_aux[cb] = aux_src + (size_t)sg * seg_stride + (slot >> 2) * 1024;
        }
        if (c_issue + 1 == n_chunks) {
            c_issue = c_start;
            seg_issue = seg_issue + 4 * NCB < seg_last ? seg_issue + 4 * NCB : seg_last;
        } else {
            c_issue = c_issue + 1;
        }
        s_issue = s_issue == 2 ? 0 : s_issue + 1;
        fetch_tables();
    }
    template <int P>
    NFL_DEV void piece() {
        if constexpr (P < MAXPW) {
            ops += 1;
            if (P == MAXPW - 1 && !i_mask) mk1 = ops;    // last piece of this chunk (unless its mask pieces follow)
            unsigned byte = (unsigned)(wave + 4 * P) * 1024u;
            const unsigned last = (unsigned)i_nbytes - 1024u;
            byte = byte < last ? byte : last;
            const unsigned vo = byte + (threadIdx.x & 63) * 16u;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(i_src + vo),
                (__attribute__((address_space(3))) void*)(i_dst + byte), 16, 0, 0);
        } else if constexpr (P < MAXP) {
            if (i_mask) {                        // uniform
                ops += 1;
                if (P == MAXP - 1) mk1 = ops;
                constexpr int cb = P - MAXPW;
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(i_aux[cb] + (unsigned)((threadIdx.x & 63) * 16u)),
                    (__attribute__((address_space(3))) void*)(i_dst + WBYTES + wave * (1024 * NCB) + cb * 1024), 16, 0, 0);
            }
        }
    }
    template <int P0, int P1>
    NFL_DEV void pieces() {
        nfl_static_for<P0, P1>([&](auto P) __attribute__((always_inline)) { piece<decltype(P)::value>(); });
    }
    NFL_DEV void prime() {
        ops = 0;
        fetch_tables();
        begin_issue();
        pieces<0, MAXP>();
        mk0 = mk1;
        begin_issue();
        pieces<0, MAXP>();
    }
#ifdef NFL_STAMPS
    unsigned long long t_wait = 0, t_bar = 0, n_cons = 0;
#endif
    NFL_DEV const char* consume() {
#ifdef NFL_STAMPS
        const unsigned long long c0 = __builtin_amdgcn_s_memtime();
#endif
        // everything up to the last piece of the oldest chunk is done; younger operations may stay in flight
        const int young = ops - mk0;
        mk0 = mk1;               // the chunk issued during the coming tile overwrites mk1 at its last piece
        // buckets on the common chunk (MAXPW weight pieces, no mask pieces) plus whole tiles of stash stores
        constexpr int B0 = MAXP - NCB;
        if (young >= B0 + 16) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(B0 + 16) : "memory");
        else if (young >= B0 + 12) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(B0 + 12) : "memory");
        else if (young >= B0 + 8) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(B0 + 8) : "memory");
        else if (young >= B0 + 4) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(B0 + 4) : "memory");
        else if (young >= B0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(B0) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#ifdef NFL_STAMPS
        const unsigned long long c1 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_barrier();
        t_wait += c1 - c0;
        t_bar += __builtin_amdgcn_s_memtime() - c1;
        n_cons += 1;
        asm volatile("" ::: "memory");
        begin_issue();
        const char* base_ = lds + s_read * SLOT_BYTES + (threadIdx.x & 63) * 16;
        s_read = s_read == 2 ? 0 : s_read + 1;
        return base_;
#endif
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        begin_issue();
        const char* base = lds + s_read * SLOT_BYTES + (threadIdx.x & 63) * 16;
        s_read = s_read == 2 ? 0 : s_read + 1;
        return base;
    }
};

// ---- stochastic rounding of the gradients (single-image modes f16 / f16w) ----------------------------------------------
// A gradient that is rounded to fp16 to the NEAREST value carries an error that is a fixed function of the value; over the
// ~1e6 samples of a step those errors do not average out of the weight-gradient sums the way independent noise would, and
// the fit's loss curve ends up below the reference's (profiles/r03_psnr_backward_attribution.txt: the round-3 measurements).
// v_cvt_sr_f16_f32 rounds up with probability = the discarded fraction (it compares the 13 discarded mantissa bits with
// bits 19..31 of its third operand: profiles/tools/sr_probe.hip), so E[rounded] = value and the errors of different samples
// are independent.  The random words come from a 24-bit LCG per lane (v_mad_u32_u24: one full-rate instruction; the product's
// bits 24..31 are folded back over the weak low bits), one step per pair-op; each conversion takes its own 13-bit window.
struct DgRng {
    unsigned s;
    NFL_DEV unsigned next() {
        s = __umul24(s, 0x9E3775u) + 0x6D2B79F5u;       // multiplier = 1 mod 4, odd increment: full period in the low 24 bits
        return s ^ (s >> 16);
    }
};
// one pair = two neighbouring features of one sample: both conversions take the same random word (the draws must be independent
// between SAMPLES -- that is what makes the sums over samples average the error out; two features of one sample may share one)
NFL_DEV unsigned dg_sr_pack(float x0, float x1, unsigned r) {
#ifdef NFL_DIAG_RN_DELTA
    return nfl_pack2<_Float16>(x0, x1);
#endif
    unsigned h;
    // the second conversion completes the register the first one wrote: the hardware wants a wait state between the two
    asm("v_cvt_sr_f16_f32 %0, %1, %3\n\ts_nop 0\n\tv_cvt_sr_f16_f32 %0, %2, %3 op_sel:[0,0,1]" : "=&v"(h) : "v"(x0), "v"(x1), "v"(r));
    return h;
}
// two pairs (the two column blocks of a pair-op): the low halves first, then the high halves, so that no conversion directly
// follows the one whose register it completes
NFL_DEV void dg_sr_pack2(float a0, float a1, unsigned ra, float b0, float b1, unsigned rb, unsigned& ha, unsigned& hb) {
#ifdef NFL_DIAG_RN_DELTA
    ha = nfl_pack2<_Float16>(a0, a1);
    hb = nfl_pack2<_Float16>(b0, b1);
    return;
#endif
    asm("v_cvt_sr_f16_f32 %0, %2, %6\n\tv_cvt_sr_f16_f32 %1, %4, %7\n\t"
        "v_cvt_sr_f16_f32 %0, %3, %6 op_sel:[0,0,1]\n\tv_cvt_sr_f16_f32 %1, %5, %7 op_sel:[0,0,1]"
        : "=&v"(ha), "=&v"(hb) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(ra), "v"(rb));
}
NFL_DEV float dg_sr_round(float x, unsigned r) {
#ifdef NFL_DIAG_RN_DELTA
    return (float)(_Float16)x;
#endif
    unsigned h;
    asm("v_cvt_sr_f16_f32 %0, %1, %2" : "=v"(h) : "v"(x), "v"(r));
    return (float)__builtin_bit_cast(_Float16, (unsigned short)h);
}

template <int NCB>
NFL_DEV void dg_zero(f16v (&acc)[NCB]) {
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
}

// epilogue of a dgrad tile, cut into 8 pair-ops: relu mask (sign of the stashed activation),
// fp16 into the next operand set and into the gradient stash
template <bool MASK, int NOUT, int NCB, int NP>
struct DgEpi {
    const f16v (&acc)[NCB];
    const unsigned (&mk)[NCB];
    h8 (&out)[NOUT][NCB][NP];
    const int ks;
    char* const (&gst)[NCB];
    const int slot;
    static constexpr int LO = NP == 2 ? NFL_GRD_SLOTS * 1024 : 0;     // the residual record follows the hi record (nfl_plan.h)
    DgRng& rng;               // per-lane generator behind the stochastic rounding of the single-image modes

    template <int OP>
    NFL_DEV void pair() {
        constexpr int s = OP / 4, j = 2 * (OP % 4);
        unsigned srh[NCB];            // single-image modes: the stochastically rounded pairs of all column blocks
        if constexpr (NP == 1) {
            // one generator step per pair-op: column block 0 takes the word, column block 1 (other samples) its other half
            const unsigned rw = rng.next();
            if constexpr (NCB == 2)
                dg_sr_pack2(acc[0][8 * s + j], acc[0][8 * s + j + 1], rw, acc[1][8 * s + j], acc[1][8 * s + j + 1],
                            __builtin_amdgcn_alignbit(rw, rw, 16), srh[0], srh[1]);
            else
                srh[0] = dg_sr_pack(acc[0][8 * s + j], acc[0][8 * s + j + 1], rw);
        }
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            unsigned hi, lo = 0u;
            if constexpr (NP == 2) {
                const float x0 = acc[cb][8 * s + j], x1 = acc[cb][8 * s + j + 1];
                float l0, l1;
                hi = nfl_split_pair<_Float16>(x0, x1, l0, l1);
                lo = nfl_pack2<_Float16>(l0, l1);
            } else {
                hi = srh[cb];
            }
            if (MASK) {
                // the forward's mask word has the pair's two predicates at bits 2*OP and 16 + 2*OP: shifted down
                // they are 0/1 per half, and an integer multiply of the gradient's fp16 bits by them masks exactly
                // (inline asm: LLVM rewrites the C form into compares and selects)
                const unsigned on = (mk[cb] >> (2 * OP)) & 0x00010001u;
                asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(hi) : "v"(hi), "v"(on));
                if constexpr (NP == 2) asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(lo) : "v"(lo), "v"(on));
            }
            reinterpret_cast<unsigned(&)[4]>(out[ks + s][cb][0])[j / 2] = hi;
            if constexpr (NP == 2) reinterpret_cast<unsigned(&)[4]>(out[ks + s][cb][NP - 1])[j / 2] = lo;
            // both stash stores of the tile are issued at its last pair-op, i.e. after every DMA piece of the
            // tile they ride in: the ring's counted vmcnt wait can then leave TWO tiles of stores outstanding
            // (the HBM write acknowledgement takes longer than one 16-MFMA row tile)
            if (OP == 7) {
                // streaming stores: the 2.6 GB of stash must not evict the weight stream from L2
                __builtin_nontemporal_store(out[ks][cb][0], reinterpret_cast<h8*>(gst[cb] + slot * 1024));
                __builtin_nontemporal_store(out[ks + 1][cb][0], reinterpret_cast<h8*>(gst[cb] + (slot + 1) * 1024));
                if constexpr (NP == 2) {
                    __builtin_nontemporal_store(out[ks][cb][NP - 1], reinterpret_cast<h8*>(gst[cb] + LO + slot * 1024));
                    __builtin_nontemporal_store(out[ks + 1][cb][NP - 1], reinterpret_cast<h8*>(gst[cb] + LO + (slot + 1) * 1024));
                }
            }
        }
    }
    template <int K, int NK>
    NFL_DEV void step() {
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

// NRT transposed row tiles (two per chunk) with up to three K segments.  TS: k-steps a tile occupies in the stream
// (more than the NK it reads when a pass leaves out the transient head's segment of the d(feat) tiles)
template <int WB, bool MASK, int NRT, int NKA, int NKB, int NKC, int NCB, class M, int TS = NKA + NKB + NKC, int NA, int NB, int NC, int NOUT, class Ring>
NFL_DEV void dg_tiles(Ring& ring, int wave_mask_off,
                      const h8 (&inA)[NA][NCB][M::NP], int ksA, const h8 (&inB)[NB][NCB][M::NP], int ksB,
                      const h8 (&inC)[NC][NCB][M::NP], int ksC,
                      h8 (&out)[NOUT][NCB][M::NP], int out_ks0, char* const (&gst)[NCB], int slot0, DgRng& rng) {
    constexpr int NK = NKA + NKB + NKC;
    constexpr int NP = M::NP, TPC = M::TPC;
    f16v acc[2][NCB];
    unsigned mk[2][NCB];
    unsigned mkq[NCB][4];       // the four mask words of the current group of tiles (arrive with its first tile)
    auto getb = [&](auto K, int cb, int part) __attribute__((always_inline)) -> const h8& {
        constexpr int k = decltype(K)::value;
        if constexpr (k < NKA) return inA[ksA + k][cb][part];
        else if constexpr (k < NKA + NKB) return inB[ksB + k - NKA][cb][part];
        else return inC[ksC + k - NKA - NKB][cb][part];
    };
    static_assert(NRT % 2 == 0, "the stream pairs the tiles of a group");
    const char* wl = nullptr;
    nfl_static_for<0, NRT>([&](auto I) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value;
        constexpr int P0 = (i % TPC) * TS;               // k-step of this tile inside its chunk
        if constexpr (i % TPC == 0) wl = ring.consume();
        dg_zero<NCB>(acc[i & 1]);
        if (MASK) {      // the slot is recycled at the next consume(): take the masks now
            if constexpr ((i & 3) == 0) {
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) {
                    typedef unsigned dg_u4 __attribute__((ext_vector_type(4)));
                    const dg_u4 v = *reinterpret_cast<const dg_u4*>(wl + WB + wave_mask_off + cb * 1024);
                    mkq[cb][0] = v[0]; mkq[cb][1] = v[1]; mkq[cb][2] = v[2]; mkq[cb][3] = v[3];
                }
            }
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) mk[i & 1][cb] = mkq[cb][i & 3];
        }
        if constexpr (i > 0) {
            DgEpi<MASK, NOUT, NCB, NP> epi{acc[(i - 1) & 1], mk[(i - 1) & 1], out, out_ks0 + 2 * (i - 1), gst, slot0 + 2 * (i - 1), rng};
            nfl_tile_p<M::PRODS, NP, NCB, NK, P0, h8>(acc[i & 1], wl, P0, getb, epi, ring);
            ring.note(2 * NCB * NP);     // the epilogue's stash stores, issued at the tile's last k-step
        } else {
            NflNoEpi epi;
            nfl_tile_p<M::PRODS, NP, NCB, NK, P0, h8>(acc[i & 1], wl, P0, getb, epi, ring);
        }
        // pieces the chunk's k-loops did not get to (piece P0 + k is issued at k-step k of its tile)
        if constexpr (TPC == 1) ring.template pieces<NK, Ring::MAXP>();
        else if constexpr ((i & 1) == 0) ring.template pieces<NK, TS>();
        else ring.template pieces<TS + NK, Ring::MAXP>();
    });
    DgEpi<MASK, NOUT, NCB, NP> last{acc[(NRT - 1) & 1], mk[(NRT - 1) & 1], out, out_ks0 + 2 * (NRT - 1), gst, slot0 + 2 * (NRT - 1), rng};
    last.all();
    ring.note(2 * NCB * NP);
}

// one tile whose rows are latent inputs: sum over the 32 samples of each segment, add to its ray's gradient
template <int NK, int NCB, class M, int NIN, class Ring>
NFL_DEV void dg_latent_tile(Ring& ring, const h8 (&in)[NIN][NCB][M::NP], int ks0, float* const (&dst)[NCB], int nvalid, int h, int c,
                            float inv_scale) {
    const char* wl = ring.consume();
    f16v acc[NCB];
    dg_zero<NCB>(acc);
    auto getb = [&](auto K, int cb, int part) __attribute__((always_inline)) -> const h8& {
        return in[ks0 + decltype(K)::value][cb][part];
    };
    NflNoEpi epi;
    nfl_tile_p<M::PRODS, M::NP, NCB, NK, 0, h8>(acc, wl, 0, getb, epi, ring);
    ring.template pieces<NK, Ring::MAXP>();
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float s = nfl_sum32(acc[cb][r]);
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            if (dst[cb] && c == 0 && row < nvalid) atomicAdd(dst[cb] + row, s * inv_scale);
        }
}

// A transposed tile whose rows are positional-encoding features (tile T covers features 32T..32T+31 of
// an N-frequency encoding): chain the feature gradients through d/dx [x, w_k sin(2^k x), w_k cos(2^k x)]
// into the gradient of the 3 encoded coordinates of this lane's sample (partial: the two lane halves
// hold different rows and are summed by the caller).
template <int N, int T, int NK, int NCB, class M, int NIN, class Ring>
NFL_DEV void dg_pe_tile(Ring& ring, const h8 (&in)[NIN][NCB][M::NP], int ks0, int h,
                        const float (&th)[NCB][3], const float (&tl)[NCB][3], const float* pw,
                        float (&g)[NCB][3]) {
    const char* wl = ring.consume();
    f16v acc[NCB];
    dg_zero<NCB>(acc);
    auto getb = [&](auto K, int cb, int part) __attribute__((always_inline)) -> const h8& {
        return in[ks0 + decltype(K)::value][cb][part];
    };
    NflNoEpi epi;
    nfl_tile_p<M::PRODS, M::NP, NCB, NK, 0, h8>(acc, wl, 0, getb, epi, ring);
    ring.template pieces<NK, Ring::MAXP>();
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f0 = 32 * T + (r & 3) + 8 * (r >> 2), f1 = f0 + 4;      // rows of lane half 0 / 1
            constexpr int NF = 6 * N + 3;
            if (f0 >= NF) continue;
            // descriptor of a feature: coordinate, scale 2^k, phase of the DERIVATIVE in turns, weight index (-1: raw x)
            const int c0 = f0 < 3 ? f0 : (f0 - 3) % 3, c1 = f1 < 3 ? f1 : (f1 - 3) % 3;
            const int k0 = f0 < 3 ? -1 : (f0 - 3) / 6, k1 = f1 < 3 ? -1 : (f1 - 3) / 6;
            const int t0 = f0 < 3 ? 0 : ((f0 - 3) % 6) / 3, t1 = f1 < 3 ? 0 : ((f1 - 3) % 6) / 3;
            const bool v1 = f1 < NF;
            float coef0 = 1.f, coef1 = v1 ? 1.f : 0.f;
            if (k0 >= 0) {
                const float sc = (float)(1 << k0);
                const float rr = __builtin_amdgcn_fractf(th[cb][c0] * sc) + tl[cb][c0] * sc + (t0 ? 0.5f : 0.25f);
                coef0 = pw[k0] * sc * nfl_sin_rev(rr);                // d sin = cos ; d cos = -sin
            }
            if (v1 && k1 >= 0) {
                const float sc = (float)(1 << k1);
                const float rr = __builtin_amdgcn_fractf(th[cb][c1] * sc) + tl[cb][c1] * sc + (t1 ? 0.5f : 0.25f);
                coef1 = pw[k1] * sc * nfl_sin_rev(rr);
            }
            const float a = acc[cb][r];
            g[cb][c0] += h ? 0.f : a * coef0;
            if (v1) g[cb][c1] += h ? a * coef1 : 0.f;
        }
}

template <int NFX, int NP, int NWP>
__global__ __launch_bounds__(256, 1) void nfl_dgrad_kernel(const DgradArgs A) {
    using M = DgMode<NP, NWP>;
    using C = NflDgradCfg<NFX, M>;
    constexpr int NKP = C::NKP, WB = C::WBYTES, NCB = C::NCB;
    constexpr int GREC = NFL_GRD_SLOTS * NP;                   // slots of a segment's gradient record: hi (+ lo)
    constexpr int GLO = NP == 2 ? NFL_GRD_SLOTS * 1024 : 0;    // byte offset of the residual record
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* const chk_lds = reinterpret_cast<int*>(smem);
    int* const aux_lds = chk_lds + NFL_MAX_CHUNKS + 8;
    const nfl_dgrad_args& a = A.a;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, c = lane & 31;
    const int N = a.n_samples, SPR = A.spr;

    const int ray0 = blockIdx.x * A.rays_per_wg;
    int ray1 = ray0 + A.rays_per_wg;
    if (ray1 > a.n_rays) ray1 = a.n_rays;
    if (ray0 >= ray1) return;
    const int seg_end = (ray1 - ray0) * SPR;
    const int ntiles = (seg_end + 4 * NCB - 1) / (4 * NCB);

    float* const pw_lds = reinterpret_cast<float*>(aux_lds + NFL_MAX_CHUNKS + 8);
    for (int i = tid; i <= A.n_chunks; i += 256) {
        chk_lds[i] = A.plan->chunk_off[i];
        aux_lds[i] = A.plan->chunk_aux[i];
    }
    if (tid < 16) pw_lds[tid] = (a.d_pe_w_xyz && tid < A.nfx_rt) ? a.d_pe_w_xyz[tid] : 1.f;
    else if (tid < 32) pw_lds[tid] = (a.d_pe_w_dir && tid - 16 < A.ndir_rt) ? a.d_pe_w_dir[tid - 16] : 1.f;
    __syncthreads();

    // loss scale of this pass (uniform; see nfl_plan.h)
    const float scale = nfl_loss_scale_from_bits(nfl_gmax_bits(a.d_gmax));
    const float inv_scale = 1.0f / scale;

    NflRingAux<C::SLOT, C::WBYTES, C::MAXP, NCB> ring;
    ring.gsrc = A.packed;
    ring.chunk_off = chk_lds;
    ring.chunk_aux = aux_lds;
    ring.lds = smem + C::LDS_TAB;
    ring.aux_src = a.d_act_stash + nfl_msk_offset((size_t)A.n_seg_total, NKP, NP);      // split activation records with NP 2
    ring.seg_stride = (size_t)NFL_MSK_WORDS * 256;      // 21 groups of 1 KiB
    ring.n_chunks = A.n_chunks;
    ring.c_start = A.c_start;
    ring.c_issue = A.c_start;
    ring.s_issue = 0;
    ring.s_read = 0;
    ring.seg_last = ray0 * SPR + seg_end - 1;
    ring.seg_issue = ray0 * SPR + wave * NCB < ring.seg_last ? ray0 * SPR + wave * NCB : ring.seg_last;
    ring.wave = wave;
    ring.lane = lane;
    ring.prime();
#ifdef NFL_STAMPS
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#endif

    // generator of the stochastic rounding: one stream per lane, mixed with the pass's gradient maximum (so the draws
    // differ from step to step) and the caller's seed (independent repetitions of a fit)
    DgRng rng{(blockIdx.x * 256u + threadIdx.x) * 0x9E3779B9u + nfl_gmax_bits(a.d_gmax) * 0x85EBCA6Bu + a.rounding_seed * 0xC2B2AE35u};
    for (int tile = 0; tile < ntiles; ++tile) {
        NflDgKArgs K = nfl_dg_kargs();      // arguments are re-read from the kernarg segment where they are used (nfl_render_impl.h: nfl_kargs)
        bool seg_ok[NCB];
        int ray[NCB];
        char* gst[NCB];
        float hg[NCB][9];
        // geometry of this lane's samples, only for the gradient w.r.t. the rays
        float xth[NCB][3], xtl[NCB][3], dth[NCB][3], dtl[NCB][3], zs[NCB], gx[NCB][3], gd[NCB][3];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const int g = tile * 4 * NCB + wave * NCB + cb;
            seg_ok[cb] = g < seg_end;
            const int gg = seg_ok[cb] ? g : seg_end - 1;
            ray[cb] = ray0 + gg / SPR;
            const int i = (gg % SPR) * 32 + c;
            const bool ok = seg_ok[cb] && i < N;
            // padded segments (zero gradients) write to a scratch record past the end: no branch in the epilogue
            gst[cb] = K->a.d_grad_stash + (size_t)(seg_ok[cb] ? ray0 * SPR + gg : K->n_seg_total) * GREC * 1024 + (2 * c + h) * 16;     // [sample][lane half][8] image, as the activation stash
            zs[cb] = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) xth[cb][k] = xtl[cb][k] = dth[cb][k] = dtl[cb][k] = gx[cb][k] = gd[cb][k] = 0.f;
            if (K->rays_tiles && K->a.d_g_rays) {
                const float* rp = K->a.d_rays + (size_t)ray[cb] * 8;
                zs[cb] = K->a.d_z[(size_t)ray[cb] * N + (i < N ? i : N - 1)];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float dk = rp[3 + k];
                    nfl_turns(rp[k] + dk * zs[cb], xth[cb][k], xtl[cb][k]);
                    nfl_turns(dk, dth[cb][k], dtl[cb][k]);
                }
            }
            const float* hp = K->a.d_head_grads + ((size_t)ray[cb] * N + (i < N ? i : N - 1)) * 9;
#pragma unroll
            for (int k = 0; k < 9; ++k) hg[cb][k] = (ok && h == 0) ? hp[k] * scale : 0.f;
            if (NP == 1) {          // the head gradients too: rounded here, so the conversions below are exact
                unsigned rw = 0u;
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    if (k % 3 == 0) rw = rng.next();
                    hg[cb][k] = dg_sr_round(hg[cb][k], __builtin_amdgcn_alignbit(rw, rw, 8 * (k % 3)));
                }
            }
        }
        // this wave's mask words in a ring slot (wl carries lane * 16 = the lane's four words of a group)
        const int moff = wave * (1024 * NCB);
        // head gradients as natural-order B operands (k = 8h + j)
        h8 dS[1][NCB][NP], dC[1][NCB][NP], dTs[1][NCB][NP], dTc[1][NCB][NP], dTb[1][NCB][NP];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const float vS[8] = {hg[cb][3], 0, 0, 0, 0, 0, 0, 0};
            const float vC[8] = {hg[cb][0], hg[cb][1], hg[cb][2], 0, 0, 0, 0, 0};
            const float vTs[8] = {hg[cb][7], 0, 0, 0, 0, 0, 0, 0};
            const float vTc[8] = {hg[cb][4], hg[cb][5], hg[cb][6], 0, 0, 0, 0, 0};
            const float vTb[8] = {hg[cb][8], 0, 0, 0, 0, 0, 0, 0};
            nfl_split8<NP>(vS, dS[0][cb]);
            nfl_split8<NP>(vC, dC[0][cb]);
            nfl_split8<NP>(vTs, dTs[0][cb]);
            nfl_split8<NP>(vTc, dTc[0][cb]);
            nfl_split8<NP>(vTb, dTb[0][cb]);
            nfl_stash8<GLO>(vS, gst[cb] + (NFL_GRD_HEADS + 0) * 1024);
            nfl_stash8<GLO>(vC, gst[cb] + (NFL_GRD_HEADS + 1) * 1024);
            if (K->use_t) {        // no weight-gradient job reads them otherwise
                nfl_stash8<GLO>(vTs, gst[cb] + (NFL_GRD_HEADS + 2) * 1024);
                nfl_stash8<GLO>(vTc, gst[cb] + (NFL_GRD_HEADS + 3) * 1024);
                nfl_stash8<GLO>(vTb, gst[cb] + (NFL_GRD_HEADS + 4) * 1024);
            }
        }
        // Two operand sets of 16 k-steps are enough for the whole walk: the transient chain ping-pongs between
        // the halves of Q, d(dir hidden) lands in Q[0..8) next to dg1 in Q[8..16), d(h8) -- straight from those two through
        // the folded W_dir' / W_t0' -- in P, then the trunk alternates Q, P, Q, ...
        h8 P[16][NCB][NP], Q[16][NCB][NP];
        K = nfl_dg_kargs();
        if (K->use_t) {
            dg_tiles<WB, true, 4, 1, 1, 1, NCB, M>(ring, moff, dTs, 0, dTc, 0, dTb, 0, Q, 0, gst, NFL_GRD_G(4), rng);
            dg_tiles<WB, true, 4, 8, 0, 0, NCB, M>(ring, moff, Q, 0, Q, 0, Q, 0, Q, 8, gst, NFL_GRD_G(3), rng);
            dg_tiles<WB, true, 4, 8, 0, 0, NCB, M>(ring, moff, Q, 8, Q, 0, Q, 0, Q, 0, gst, NFL_GRD_G(2), rng);
            dg_tiles<WB, true, 4, 8, 0, 0, NCB, M>(ring, moff, Q, 0, Q, 0, Q, 0, Q, 8, gst, NFL_GRD_G(1), rng);
            float* gt[NCB];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const size_t row = K->a.d_latent_row ? (size_t)K->a.d_latent_row[ray[cb]] : (size_t)ray[cb];     // table row or ray
                gt[cb] = (K->a.d_g_t_emb && seg_ok[cb]) ? K->a.d_g_t_emb + row * K->n_tau : nullptr;
            }
            dg_latent_tile<8, NCB, M>(ring, Q, 8, gt, K->n_tau, h, c, inv_scale);
        }
        dg_tiles<WB, true, 4, 1, 0, 0, NCB, M>(ring, moff, dC, 0, dC, 0, dC, 0, Q, 0, gst, NFL_GRD_DIRH, rng);
        K = nfl_dg_kargs();
        if (K->has_a) {
            float* ga[NCB];
            float* ga2[NCB];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const size_t row = K->a.d_latent_row ? (size_t)K->a.d_latent_row[ray[cb]] : (size_t)ray[cb];
                ga[cb] = (K->a.d_g_a_emb && seg_ok[cb]) ? K->a.d_g_a_emb + row * K->n_a : nullptr;
                ga2[cb] = ga[cb] ? ga[cb] + 32 : nullptr;
            }
            dg_latent_tile<8, NCB, M>(ring, Q, 0, ga, K->n_a < 32 ? K->n_a : 32, h, c, inv_scale);
            dg_latent_tile<8, NCB, M>(ring, Q, 0, ga2, K->n_a - 32, h, c, inv_scale);        // <= 0 rows for codes of <= 32
        }
        K = nfl_dg_kargs();
        if (K->rays_tiles) dg_pe_tile<4, 0, 8, NCB, M>(ring, Q, 0, h, dth, dtl, pw_lds + 16, gd);
        // d(h8) straight from the 128-wide head gradients: xyz_encoding_final is folded into W_dir' / W_t0' (nfl_plan.cpp),
        // so there are no d(feat) tiles; tile = [W_dir'^T: 8 k-steps | W_t0'^T: 8 (fields with a transient head) | W_sigma^T: 1]
        if (K->use_t) {
            dg_tiles<WB, true, 8, 8, 8, 1, NCB, M>(ring, moff, Q, 0, Q, 8, dS, 0, P, 0, gst, NFL_GRD_D(8), rng);
        } else if (K->has_t) {      // the stream carries the transient segment: multiply it by zeros
#pragma unroll
            for (int ks = 8; ks < 16; ++ks)
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        Q[ks][cb][0][j] = (_Float16)0.f;
                        Q[ks][cb][NP - 1][j] = (_Float16)0.f;
                    }
            dg_tiles<WB, true, 8, 8, 8, 1, NCB, M>(ring, moff, Q, 0, Q, 8, dS, 0, P, 0, gst, NFL_GRD_D(8), rng);
        } else {
            dg_tiles<WB, true, 8, 8, 1, 0, NCB, M>(ring, moff, Q, 0, dS, 0, dS, 0, P, 0, gst, NFL_GRD_D(8), rng);
        }
        dg_tiles<WB, true, 8, 16, 0, 0, NCB, M>(ring, moff, P, 0, P, 0, P, 0, Q, 0, gst, NFL_GRD_D(7), rng);
        dg_tiles<WB, true, 8, 16, 0, 0, NCB, M>(ring, moff, Q, 0, Q, 0, Q, 0, P, 0, gst, NFL_GRD_D(6), rng);
        dg_tiles<WB, true, 8, 16, 0, 0, NCB, M>(ring, moff, P, 0, P, 0, P, 0, Q, 0, gst, NFL_GRD_D(5), rng);
        dg_tiles<WB, true, 8, 16, 0, 0, NCB, M>(ring, moff, Q, 0, Q, 0, Q, 0, P, 0, gst, NFL_GRD_D(4), rng);
        K = nfl_dg_kargs();
        if (K->rays_tiles) {       // skip connection: delta_5 (still in Q) reaches the encoded position too
            dg_pe_tile<NFX, 0, 16, NCB, M>(ring, Q, 0, h, xth, xtl, pw_lds, gx);
            dg_pe_tile<NFX, 1, 16, NCB, M>(ring, Q, 0, h, xth, xtl, pw_lds, gx);
            if (NKP > 4) dg_pe_tile<NFX, 2, 16, NCB, M>(ring, Q, 0, h, xth, xtl, pw_lds, gx);
        }
        dg_tiles<WB, true, 8, 16, 0, 0, NCB, M>(ring, moff, P, 0, P, 0, P, 0, Q, 0, gst, NFL_GRD_D(3), rng);
        dg_tiles<WB, true, 8, 16, 0, 0, NCB, M>(ring, moff, Q, 0, Q, 0, Q, 0, P, 0, gst, NFL_GRD_D(2), rng);
        dg_tiles<WB, true, 8, 16, 0, 0, NCB, M>(ring, moff, P, 0, P, 0, P, 0, Q, 0, gst, NFL_GRD_D(1), rng);
        if (K->rays_tiles) {
            dg_pe_tile<NFX, 0, 16, NCB, M>(ring, Q, 0, h, xth, xtl, pw_lds, gx);
            dg_pe_tile<NFX, 1, 16, NCB, M>(ring, Q, 0, h, xth, xtl, pw_lds, gx);
            if (NKP > 4) dg_pe_tile<NFX, 2, 16, NCB, M>(ring, Q, 0, h, xth, xtl, pw_lds, gx);
            if (K->a.d_g_rays) {
                // x = o + d z ; the view direction is d itself (no caller passes view_dir with learnable poses)
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const float gxk = gx[cb][k] + __shfl_xor(gx[cb][k], 32);
                        // with a separate view_dir the direction encoding is data (rendering.py:236-238): no share for rays_d
                        const float gdk = K->a.dir_is_data ? 0.f : gd[cb][k] + __shfl_xor(gd[cb][k], 32);
                        const float so = nfl_sum32(gxk) * inv_scale;
                        const float sd = nfl_sum32(gxk * zs[cb] + gdk) * inv_scale;
                        if (lane == 0 && seg_ok[cb]) {
                            atomicAdd(K->a.d_g_rays + (size_t)ray[cb] * 8 + k, so);
                            atomicAdd(K->a.d_g_rays + (size_t)ray[cb] * 8 + 3 + k, sd);
                        }
                    }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef NFL_STAMPS
    if (lane == 0 && blockIdx.x < 1024) {
        unsigned long long* o = nfl_stamp_buf + (blockIdx.x * 4 + wave) * NFL_NSTAMP;
        o[0] = __builtin_amdgcn_s_memtime() - t_begin;
        o[1] = ring.t_wait;
        o[2] = ring.t_bar;
        o[3] = ring.n_cons;
        o[4] = ntiles;
    }
#endif
}

template <int NFX, int NP, int NWP>
static int launch_dgrad(const NflPlan* hp, const void* d_plan, const void* d_packed, const nfl_dgrad_args* args,
                        hipStream_t stream) {
    using C = NflDgradCfg<NFX, DgMode<NP, NWP>>;
    constexpr int NCB_ = C::NCB;
    DgradArgs A;
    A.plan = static_cast<const NflPlan*>(d_plan);
    A.packed = static_cast<const char*>(d_packed);
    A.a = *args;
    A.has_a = hp->has_a;
    A.has_t = hp->has_t;
    A.use_t = (hp->has_t && args->use_transient) ? 1 : 0;
    A.n_chunks = hp->n_chunks;
    A.c_start = (hp->has_t && !A.use_t) ? hp->n_chunks_sigma : 0;     // skip the transient head's chunks
    if (args->d_g_rays && (!A.rays_tiles || !args->d_rays || !args->d_z)) return NFL_EINVAL;
    A.spr = (args->n_samples + 31) / 32;
    A.nkp = hp->nkp;
    A.n_seg_total = args->n_rays * A.spr;
    A.rays_tiles = hp->reserved_flags & 1;
    A.nfx_rt = hp->n_emb_xyz;
    A.ndir_rt = (hp->reserved_flags >> 8) & 0xff;
    A.n_a = hp->n_a;
    A.n_tau = hp->n_tau;
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    int rpw = (args->n_rays + ncu - 1) / ncu;
    const int rays_per_tile = 4 * NCB_ / A.spr;
    if (rays_per_tile > 1) rpw = (rpw + rays_per_tile - 1) / rays_per_tile * rays_per_tile;
    if (rpw < 1) rpw = 1;
    A.rays_per_wg = rpw;
    const int grid = (args->n_rays + rpw - 1) / rpw;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&nfl_dgrad_kernel<NFX, NP, NWP>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES) != hipSuccess)
            return NFL_ENODEV;
        attr_set = true;
    }
    hipLaunchKernelGGL((nfl_dgrad_kernel<NFX, NP, NWP>), dim3(grid), dim3(256), C::LDS_BYTES, stream, A);
    return hipGetLastError() == hipSuccess ? NFL_OK : NFL_ELAUNCH;
}

// Compiled twice (Makefile): NFL_DGRAD_WIDE=0 -> nfl_dgrad.o, the C entry point and the kernels for up to 10 frequencies;
// NFL_DGRAD_WIDE=1 -> nfl_dgrad_w.o, the kernels for 11..15 -- the two halves compile in parallel.
#ifndef NFL_DGRAD_WIDE
#define NFL_DGRAD_WIDE 0
#endif
#if NFL_DGRAD_WIDE
extern "C" int nfl_mlp_dgrad_wide(const NflPlan* hp, const void* d_bwd_plan, const void* d_bwd_packed, const nfl_dgrad_args* args,
                                  void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hp->prec == NFL_PREC_F16 && hp->nsplit == 1) return launch_dgrad<15, 1, 1>(hp, d_bwd_plan, d_bwd_packed, args, s);
    if (hp->prec == NFL_PREC_F16W && hp->nsplit == 3) return launch_dgrad<15, 1, 2>(hp, d_bwd_plan, d_bwd_packed, args, s);
    if (hp->prec == NFL_PREC_F16X3 && hp->nsplit == 3) return launch_dgrad<15, 2, 2>(hp, d_bwd_plan, d_bwd_packed, args, s);
    return NFL_EINVAL;
}
#else
extern "C" int nfl_mlp_dgrad_wide(const NflPlan*, const void*, const void*, const nfl_dgrad_args*, void*);

extern "C" int nfl_mlp_dgrad(const void* h_bwd_plan, const void* d_bwd_plan, const void* d_bwd_packed,
                             const nfl_dgrad_args* args, void* stream) {
    const NflPlan* hp = static_cast<const NflPlan*>(h_bwd_plan);
    if (!hp || hp->magic != NFL_PLAN_MAGIC || !hp->is_bwd || !d_bwd_plan || !d_bwd_packed || !args) return NFL_EINVAL;
    if (!args->d_head_grads || !args->d_act_stash || !args->d_grad_stash || !args->d_gmax) return NFL_EINVAL;
    if (args->n_rays < 0 || args->n_samples < 1) return NFL_EINVAL;
    if (args->n_rays == 0) return NFL_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hp->n_emb_xyz < 1 || hp->n_emb_xyz > NFL_MAX_EMB_XYZ) return NFL_EINVAL;
    if (hp->n_emb_xyz > 10) return nfl_mlp_dgrad_wide(hp, d_bwd_plan, d_bwd_packed, args, stream);      // encoder widths: nfl_plan.h
    if (hp->prec == NFL_PREC_F16 && hp->nsplit == 1) return launch_dgrad<10, 1, 1>(hp, d_bwd_plan, d_bwd_packed, args, s);
    if (hp->prec == NFL_PREC_F16W && hp->nsplit == 3) return launch_dgrad<10, 1, 2>(hp, d_bwd_plan, d_bwd_packed, args, s);
    if (hp->prec == NFL_PREC_F16X3 && hp->nsplit == 3) return launch_dgrad<10, 2, 2>(hp, d_bwd_plan, d_bwd_packed, args, s);
    return NFL_EINVAL;
}
#endif
