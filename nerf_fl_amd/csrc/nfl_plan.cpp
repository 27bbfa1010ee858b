// nfl_plan.cpp -- host-side builder of the packed-stream schedule (see nfl_plan.h).
#include "nfl_plan.h"

#include <string.h>

#include "../../include/nerf_fl_amd.h"

namespace {

struct Builder {
    NflPlan* p;
    int ks_cursor = 0;
    int chunk_ks = 0;     // k-steps in the currently open chunk
    bool chunk_open = false;

    void begin_chunk() {
        p->chunk_off[p->n_chunks] = ks_cursor * p->ks_bytes;
        p->n_chunks++;
        chunk_ks = 0;
        chunk_open = true;
    }
    void end_chunk() {
        if (chunk_ks > p->max_chunk_ks) p->max_chunk_ks = chunk_ks;
        chunk_open = false;
    }
    NflRowTile& add_tile() {
        NflRowTile& t = p->rt[p->n_rt++];
        memset(&t, 0, sizeof(t));
        t.frag_off = ks_cursor;
        return t;
    }
    void close_tile(NflRowTile& t) {
        int n = 0;
        for (int s = 0; s < t.nseg; ++s) n += t.seg[s].nks;
        t.nks = n;
        ks_cursor += n;
        chunk_ks += n;
    }
    static void blk(NflRowTile& t, int layer, int nrows, int src_row0, int dst_row) {
        NflBlk& b = t.blk[t.nblk++];
        b.layer = (int16_t)layer; b.nrows = (int16_t)nrows; b.src_row0 = (int16_t)src_row0; b.dst_row = (int16_t)dst_row;
    }
    static void seg(NflRowTile& t, int nks, int kind, int col0, int ncols, int layer = -1) {
        NflSeg& s = t.seg[t.nseg++];
        s.nks = (int16_t)nks; s.kind = (int16_t)kind; s.col0 = (int16_t)col0; s.ncols = (int16_t)ncols;
        s.layer = (int16_t)layer;
    }
    // one transposed (dgrad) tile: rows = columns [tcol0, tcol0+tncols) of the sources.  `join`: append it to the chunk
    // of the previous tile (the kernel walks its 4- and 8-tile groups two tiles per chunk: one barrier per pair)
    template <class SegFn>
    void ttile(int tcol0, int tncols, int aux, SegFn segs, bool join = false) {
        if (!join) {
            p->chunk_aux[p->n_chunks] = aux;
            begin_chunk();
        }
        NflRowTile& t = add_tile();
        t.trans = 1; t.tcol0 = (int16_t)tcol0; t.tncols = (int16_t)tncols;
        segs(t);
        close_tile(t);
        end_chunk();
    }
    // a dense layer of `rows` outputs split into 32-row tiles, `tiles_per_chunk` tiles per chunk
    template <class SegFn>
    void layer(int layer_id, int rows, int tiles_per_chunk, SegFn segs) {
        int ntile = (rows + 31) / 32;
        for (int i = 0; i < ntile; ++i) {
            if (i % tiles_per_chunk == 0) begin_chunk();
            NflRowTile& t = add_tile();
            int n = rows - 32 * i < 32 ? rows - 32 * i : 32;
            blk(t, layer_id, n, 32 * i, 0);
            segs(t);
            close_tile(t);
            if (i % tiles_per_chunk == tiles_per_chunk - 1 || i == ntile - 1) end_chunk();
        }
    }
};

}  // namespace

static void common_init(const nfl_field_desc* d, int prec, NflPlan* p) {
    memset(p, 0, sizeof(*p));
    p->magic = NFL_PLAN_MAGIC;
    p->prec = prec;
    p->nsplit = prec == NFL_PREC_F16X3 ? 3 : 1;
    p->ks_bytes = prec == NFL_PREC_F16X3 ? 2048 : 1024;
    p->n_emb_xyz = d->n_emb_xyz;
    const int cx = 6 * d->n_emb_xyz + 3, cd = 6 * d->n_emb_dir + 3;
    p->nkp = nfl_nkp_for(d->n_emb_xyz);
    p->has_a = d->encode_appearance ? 1 : 0;
    p->has_t = d->encode_transient ? 1 : 0;
    p->n_a = p->has_a ? d->n_a : 0;
    p->n_tau = d->n_tau;
    p->beta_min = d->beta_min;
    const int W = NFL_W, H = NFL_W / 2;
    for (int i = 0; i < 8; ++i) p->ld[NFL_P_XYZ1 + i] = i == 0 ? cx : (i == 4 ? W + cx : W);
    p->ld[NFL_P_FINAL] = W;
    p->ld[NFL_P_DIR] = W + cd + p->n_a;
    p->ld[NFL_P_SIGMA] = W;
    p->ld[NFL_P_RGB] = H;
    p->ld[NFL_P_T0] = W + d->n_tau;
    p->ld[NFL_P_T0 + 1] = p->ld[NFL_P_T0 + 2] = p->ld[NFL_P_T0 + 3] = H;
    p->ld[NFL_P_TSIGMA] = p->ld[NFL_P_TRGB] = p->ld[NFL_P_TBETA] = H;
    for (int i = 0; i <= NFL_MAX_CHUNKS; ++i) p->chunk_aux[i] = -1;
}

static int check_desc(const nfl_field_desc* d) {
    if (!d) return NFL_EINVAL;
    if (d->n_emb_xyz < 1 || d->n_emb_xyz > NFL_MAX_EMB_XYZ) return NFL_EINVAL;
    if (d->n_emb_dir < 1 || d->n_emb_dir > NFL_MAX_EMB_DIR) return NFL_EINVAL;
    // latent widths (opt.py: --N_a 48, --N_tau 16): narrower codes run in the same 3 / 1 k-steps on zero-padded weight columns
    if (d->encode_appearance && (d->n_a < 1 || d->n_a > 48)) return NFL_EINVAL;
    if (d->encode_transient && (d->n_tau < 1 || d->n_tau > 16)) return NFL_EINVAL;
    return NFL_OK;
}

// The dgrad stream (fp16): one transposed row tile per chunk, in the order the
// backward kernel walks the network (heads first).  chunk_aux names the mask word (nfl_msk_*)
// that holds the relu mask of that tile.
extern "C" int nfl_plan_fill_bwd(const nfl_field_desc* d, int rays_grad, int bwd_prec, NflPlan* p) {
    if (!p || check_desc(d) != NFL_OK) return NFL_EINVAL;
    if (bwd_prec != NFL_PREC_F16 && bwd_prec != NFL_PREC_F16W && bwd_prec != NFL_PREC_F16X3) return NFL_EINVAL;
    // fp16 fragments of the transposed weights, gradients loss-scaled (nfl_loss_scale_from_bits).  NFL_PREC_F16: hi
    // fragments only (1 KiB per k-step), the 4- and 8-tile groups two tiles per ring chunk (one barrier per pair).
    // NFL_PREC_F16W / NFL_PREC_F16X3: hi + lo fragments (2 KiB per k-step: the chain sees the weights to fp32 class), one
    // tile per chunk (a pair would not fit a ring slot).  `prec` records which dgrad kernel reads the stream.
    common_init(d, bwd_prec == NFL_PREC_F16 ? NFL_PREC_F16 : NFL_PREC_F16X3, p);
    p->prec = bwd_prec;
    const bool pair = bwd_prec == NFL_PREC_F16;
    p->elem = 0;
    p->is_bwd = 1;
    const int cx = 6 * d->n_emb_xyz + 3, cd = 6 * d->n_emb_dir + 3;
    const int W = NFL_W, H = NFL_W / 2, nkp = p->nkp;
    Builder b{p};
    if (p->has_t) {
        for (int t = 0; t < 4; ++t)
            b.ttile(32 * t, 32, nfl_msk_g(4) + t, [&](NflRowTile& r) {
                Builder::seg(r, 1, NFL_SEG_NAT, 0, 1, NFL_P_TSIGMA);
                Builder::seg(r, 1, NFL_SEG_NAT, 0, 3, NFL_P_TRGB);
                Builder::seg(r, 1, NFL_SEG_NAT, 0, 1, NFL_P_TBETA);
            }, pair && (t & 1));
        for (int j = 3; j >= 1; --j)
            for (int t = 0; t < 4; ++t)
                b.ttile(32 * t, 32, nfl_msk_g(j) + t,
                        [&](NflRowTile& r) { Builder::seg(r, 8, NFL_SEG_ACT, 0, H, NFL_P_T0 + j); }, pair && (t & 1));
        b.ttile(W, d->n_tau, -1, [&](NflRowTile& r) { Builder::seg(r, 8, NFL_SEG_ACT, 0, H, NFL_P_T0); });
    }
    const int first_static_chunk = p->n_chunks;      // the backward of a pass without the transient head starts here
    for (int t = 0; t < 4; ++t)
        b.ttile(32 * t, 32, nfl_msk_dirh() + t,
                [&](NflRowTile& r) { Builder::seg(r, 1, NFL_SEG_NAT, 0, 3, NFL_P_RGB); }, pair && (t & 1));
    if (p->has_a)
        for (int t = 0; t < 2; ++t)
            b.ttile(W + cd + 32 * t, t == 0 ? (p->n_a < 32 ? p->n_a : 32) : (p->n_a > 32 ? p->n_a - 32 : 0), -1,
                    [&](NflRowTile& r) { Builder::seg(r, 8, NFL_SEG_ACT, 0, H, NFL_P_DIR); });
    if (rays_grad)      // rows of W_dir^T that multiply the encoded view direction
        b.ttile(W, cd, -1, [&](NflRowTile& r) { Builder::seg(r, 8, NFL_SEG_ACT, 0, H, NFL_P_DIR); });
    // d(h8) = W_dir'^T d(dirh) (+ W_t0'^T d(g1)) + W_sigma^T d(sigma), W' the weights folded through xyz_encoding_final
    for (int t = 0; t < 8; ++t)
        b.ttile(32 * t, 32, nfl_msk_h(8) + t, [&](NflRowTile& r) {
            Builder::seg(r, 8, NFL_SEG_ACT, 0, H, NFL_P_DIR);
            if (p->has_t) Builder::seg(r, 8, NFL_SEG_ACT, 0, H, NFL_P_T0);
            Builder::seg(r, 1, NFL_SEG_NAT, 0, 1, NFL_P_SIGMA);
        }, pair && (t & 1));
    const int npe = (nkp + 1) / 2;          // 32-row tiles covering the encoded position AS THE KERNEL WALKS IT (its instantiation's width; rows beyond cx are zero)
    for (int l = 8; l >= 2; --l) {          // layer l (1-based) transposed -> gradient of h_{l-1}
        for (int t = 0; t < 8; ++t)
            b.ttile((l == 5 ? cx : 0) + 32 * t, 32, nfl_msk_h(l - 1) + t,
                    [&](NflRowTile& r) { Builder::seg(r, 16, NFL_SEG_ACT, 0, W, NFL_P_XYZ1 + l - 1); }, pair && (t & 1));
        if (l == 5 && rays_grad)            // skip connection: rows that multiply the encoded position
            for (int t = 0; t < npe; ++t)
                b.ttile(32 * t, cx - 32 * t < 32 ? (cx - 32 * t > 0 ? cx - 32 * t : 0) : 32, -1,
                        [&](NflRowTile& r) { Builder::seg(r, 16, NFL_SEG_ACT, 0, W, NFL_P_XYZ1 + 4); });
    }
    if (rays_grad)                          // layer 1 transposed -> gradient of the encoded position
        for (int t = 0; t < npe; ++t)
            b.ttile(32 * t, cx - 32 * t < 32 ? (cx - 32 * t > 0 ? cx - 32 * t : 0) : 32, -1,
                    [&](NflRowTile& r) { Builder::seg(r, 16, NFL_SEG_ACT, 0, W, NFL_P_XYZ1); });
    p->reserved_flags = (rays_grad ? 1 : 0) | (d->n_emb_dir << 8);
    p->n_rt_sigma = p->n_rt_static = p->n_rt;
    p->n_chunks_static = p->n_chunks;
    p->n_chunks_sigma = first_static_chunk;          // (field re-used by the dgrad stream)
    p->total_ks = b.ks_cursor;
    p->chunk_off[p->n_chunks] = p->total_ks * p->ks_bytes;
    p->stream_bytes = p->total_ks * p->ks_bytes;
    p->bias_off = p->stream_bytes;
    p->packed_bytes = p->bias_off + p->n_rt * 32 * 4;
    if (p->n_rt > NFL_MAX_RT || p->n_chunks > NFL_MAX_CHUNKS) return NFL_EINVAL;
    return NFL_OK;
}

extern "C" int nfl_plan_fill(const nfl_field_desc* d, int prec, NflPlan* p) {
    if (!p || check_desc(d) != NFL_OK) return NFL_EINVAL;
    if (prec != NFL_PREC_F16X3 && prec != NFL_PREC_F16) return NFL_EINVAL;
    common_init(d, prec, p);
    const int cx = 6 * d->n_emb_xyz + 3, cd = 6 * d->n_emb_dir + 3;
    const int W = NFL_W, H = NFL_W / 2, nkp = p->nkp;

    Builder b{p};
    // trunk
    b.layer(NFL_P_XYZ1 + 0, W, 2, [&](NflRowTile& t) { Builder::seg(t, nkp, NFL_SEG_NAT, 0, cx); });
    for (int i = 1; i < 8; ++i) {
        if (i == 4)
            b.layer(NFL_P_XYZ1 + i, W, 1, [&](NflRowTile& t) {
                Builder::seg(t, nkp, NFL_SEG_NAT, 0, cx);
                Builder::seg(t, 16, NFL_SEG_ACT, cx, W);
            });
        else
            b.layer(NFL_P_XYZ1 + i, W, 1, [&](NflRowTile& t) { Builder::seg(t, 16, NFL_SEG_ACT, 0, W); });
    }
    b.layer(NFL_P_SIGMA, 1, 1, [&](NflRowTile& t) { Builder::seg(t, 16, NFL_SEG_ACT, 0, W); });
    p->n_rt_sigma = p->n_rt;
    p->n_chunks_sigma = p->n_chunks;
    // static head.  xyz_encoding_final is linear and feeds dir_encoding.0 / transient_encoding.0 only: it is folded into
    // their first 256 input columns (W' = W[:, :256] W_fin, b' = b + W[:, :256] b_fin: nfl_compose_forward writes the
    // folded copies the packer reads), so the stream has no tiles for it and both layers read h8 directly
    b.layer(NFL_P_DIR, H, 1, [&](NflRowTile& t) {
        Builder::seg(t, 16, NFL_SEG_ACT, 0, W);
        Builder::seg(t, 2, NFL_SEG_NAT, W, cd);
        if (p->has_a) Builder::seg(t, 3, NFL_SEG_NAT, W + cd, p->n_a);
    });
    b.layer(NFL_P_RGB, 3, 1, [&](NflRowTile& t) { Builder::seg(t, 8, NFL_SEG_ACT, 0, H); });
    p->n_rt_static = p->n_rt;
    p->n_chunks_static = p->n_chunks;
    // transient head
    if (p->has_t) {
        b.layer(NFL_P_T0, H, 1, [&](NflRowTile& t) {
            Builder::seg(t, 16, NFL_SEG_ACT, 0, W);
            Builder::seg(t, 1, NFL_SEG_NAT, W, d->n_tau);
        });
        for (int j = 1; j < 4; ++j)
            b.layer(NFL_P_T0 + j, H, 2, [&](NflRowTile& t) { Builder::seg(t, 8, NFL_SEG_ACT, 0, H); });
        b.begin_chunk();
        NflRowTile& t = b.add_tile();
        Builder::blk(t, NFL_P_TSIGMA, 1, 0, 0);
        Builder::blk(t, NFL_P_TRGB, 3, 0, 1);
        Builder::blk(t, NFL_P_TBETA, 1, 0, 8);
        Builder::seg(t, 8, NFL_SEG_ACT, 0, H);
        b.close_tile(t);
        b.end_chunk();
    }
    p->total_ks = b.ks_cursor;
    p->chunk_off[p->n_chunks] = p->total_ks * p->ks_bytes;
    p->stream_bytes = p->total_ks * p->ks_bytes;
    p->bias_off = p->stream_bytes;
    p->packed_bytes = p->bias_off + p->n_rt * 32 * 4;
    if (p->n_rt > NFL_MAX_RT || p->n_chunks > NFL_MAX_CHUNKS) return NFL_EINVAL;
    return NFL_OK;
}
