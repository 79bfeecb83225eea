// Training step of the encoder: forward that keeps what the backward needs, and the backward
// itself (the gradients torch autograd would produce for pretraining.py:94-134), as plain
// sequences of kernel launches on one stream.  All saved activations and temporaries live in
// one caller-owned workspace that must stay untouched between the two calls.
#include "common.h"

namespace {

inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

struct LayerSaved {
    float *x_in;     // residual stream entering the layer            f32 [M,E]
    void *h1;        // LayerNorm 1 output                            dtype [M,E]
    void *qkv;       // scaled / rotated q, rotated k, v              dtype [M,3E]
    void *ao;        // attention output before out_proj              dtype [M,E]
    float *lse;      // softmax log-sum-exp                           f32 [B,H,T]
    float *x_mid;    // residual stream after the attention block     f32 [M,E]
    void *h2;        // LayerNorm 2 output                            dtype [M,E]
    void *u;         // fc1 pre-activation                            dtype [M,F]
    void *mid;       // gelu(u)                                       dtype [M,F]
};

constexpr int kMaxLayers = 64;

struct TrainWs {
    LayerSaved L[kMaxLayers];
    float *x;            // running residual stream / its gradient     f32 [M,E]
    void *hf;            // final LayerNorm output                      dtype [M,E]
    void *u_lm, *g_lm;   // LM head dense pre-activation and gelu       dtype [M,E]
    void *u_st, *g_st;   // structure head
    uint8_t *pad;
    uint64_t *words;
    // backward temporaries
    float *t0, *t1;      // f32 [M,E]
    void *dyb;           // dtype [M,E]
    void *wide;          // dtype [M,max(F + 64,3E)]; F-wide contents use rows ld_f elements apart
    int ld_f;            // row stride of every [M,F] activation: F + 64
    void *tr_a;          // dtype [max(F,3E), Mp]
    void *tr_b;          // dtype [max(F,E), Mp]
    char *wt_all;        // dtype, every weight matrix transposed (one launch per backward pass): wt_of()
    float *slabs;        // split-K partial weight gradients
    float *delta;        // f32 [B,H,T]
    size_t slab_floats;
    size_t total;
};

// Token-dimension split of a weight-gradient product.  bf16 with 256-divisible shapes runs the
// 256 x 256 kernel, one workgroup per CU: as many splits as fill the chip once.  Otherwise
// 128 x 128 tiles, about four workgroups per CU.
int wgrad_splits(int dt, int n_out, int n_in, int nk) {
    long s;
    if (dt == TC_BF16 && n_out % 256 == 0 && n_in % 256 == 0) {
        const long tiles = (long)(n_out / 256) * (n_in / 256);
        s = 256 / tiles;
    } else {
        const long tiles = (long)((n_out + 127) / 128) * (n_in / 128);
        s = 1024 / (tiles > 0 ? tiles : 1);
    }
    if (s < 1) s = 1;
    if (s > 64) s = 64;
    if (s > nk) s = nk;
    return (int)s;
}

TrainWs carve(const tc_encoder_desc *d, int B, int T, char *base) {
    const size_t M = (size_t)B * T, E = d->E, F = d->F;
    const size_t esz = d->dtype == TC_BF16 ? 2 : 4;
    const size_t Mp = (M + 63) / 64 * 64;
    // [M,F] activations keep their rows 128 bytes further apart than F elements: a row stride of
    // 6144 bytes (F = 3072, bf16) slows the stores of the GEMM epilogues that write them.
    const size_t ldf = F + 64;
    const size_t wide = ldf > 3 * E ? ldf : 3 * E;
    const int nk = (int)(Mp / (128 / esz));
    TrainWs w{};
    w.ld_f = (int)ldf;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char *p = base ? base + off : nullptr;
        off += align_up(bytes);
        return p;
    };
    for (int l = 0; l < d->num_layers && l < kMaxLayers; ++l) {
        LayerSaved &s = w.L[l];
        s.x_in = reinterpret_cast<float *>(take(M * E * 4));
        s.h1 = take(M * E * esz);
        s.qkv = take(M * 3 * E * esz);
        s.ao = take(M * E * esz);
        s.lse = reinterpret_cast<float *>(take((size_t)B * d->H * T * 4));
        s.x_mid = reinterpret_cast<float *>(take(M * E * 4));
        s.h2 = take(M * E * esz);
        s.u = take(M * ldf * esz);
        s.mid = take(M * ldf * esz);
    }
    w.x = reinterpret_cast<float *>(take(M * E * 4));
    w.hf = take(M * E * esz);
    w.u_lm = take(M * E * esz);
    w.g_lm = take(M * E * esz);
    w.u_st = take(M * E * esz);
    w.g_st = take(M * E * esz);
    w.pad = reinterpret_cast<uint8_t *>(take(M));
    w.words = reinterpret_cast<uint64_t *>(take((size_t)B * ((T + 63) / 64) * 8));
    w.t0 = reinterpret_cast<float *>(take(M * E * 4));
    w.t1 = reinterpret_cast<float *>(take(M * E * 4));
    w.dyb = take(M * E * esz);
    w.wide = take(M * wide * esz);
    w.tr_a = take(wide * Mp * esz);
    w.tr_b = take((F > E ? F : E) * Mp * esz);
    w.wt_all = take(((size_t)d->num_layers * (4 * E * E + 2 * E * F) + 2 * E * E) * esz);
    size_t sf = 0;
    const size_t shapes[4][2] = {{E, F}, {F, E}, {3 * E, E}, {E, E}};
    for (auto &sh : shapes) {
        const size_t v = (size_t)wgrad_splits(d->dtype, (int)sh[0], (int)sh[1], nk) * sh[0] * sh[1];
        if (v > sf) sf = v;
    }
    w.slab_floats = sf;
    w.slabs = reinterpret_cast<float *>(take(sf * 4));
    w.delta = reinterpret_cast<float *>(take((size_t)B * d->H * T * 4));
    w.total = off;
    return w;
}

int check_desc(const tc_encoder_desc *d, const char *who) {
    TC_REQUIRE(d != nullptr, "%s: null descriptor", who);
    TC_REQUIRE(d->dtype == TC_BF16 || d->dtype == TC_F32, "%s: bad dtype %d", who, d->dtype);
    TC_REQUIRE(d->num_layers > 0 && d->num_layers <= kMaxLayers && d->layers, "%s: bad layer count", who);
    TC_REQUIRE(d->E > 0 && d->H > 0 && d->E == d->H * 64, "%s: E=%d H=%d needs head_dim 64", who, d->E, d->H);
    TC_REQUIRE(d->E % 128 == 0 && d->F % 128 == 0 && d->E <= 768, "%s: E=%d (<= 768) and F=%d must be multiples of 128",
               who, d->E, d->F);
    TC_REQUIRE((d->vocab == 10) && (d->n_struct == 3), "%s: the head gradients are built for 10 and 3 classes", who);
    TC_REQUIRE(d->tok_table && d->species_table && d->lnf_g && d->lnf_b && d->w_heads && d->b_heads && d->lm_ln_g &&
                   d->lm_ln_b && d->lm_bias && d->st_ln_g && d->st_ln_b && d->st_w && d->st_bias,
               "%s: null weight pointer", who);
    return TC_OK;
}

#define TC_TRY(expr)                    \
    do {                                \
        int rc__ = (expr);              \
        if (rc__ != TC_OK) return rc__; \
    } while (0)

// dW[n_out, n_in] += dY^T X with dY [M, n_out] (element type in_dt) and X [M, n_in] (dtype dt);
// dbias[n_out] += column sums of dY (bf16 mode: skipped when dbias is null).  dy_rowmajor (optional, dense) receives dY converted to dt.
// ld_dy / ld_x: row strides of dY and X in elements.
int linear_wgrad(const tc_encoder_desc *d, TrainWs &w, int in_dt, const void *dy, int ld_dy, void *dy_rowmajor,
                 const void *x, int ld_x, int M, int n_out, int n_in, float *dW, float *dbias, void *stream) {
    const int dt = d->dtype;
    const int esz = dt == TC_BF16 ? 2 : 4;
    const int Mp = (M + 63) / 64 * 64;
    const int nk = Mp / (128 / esz);
    const int splits = wgrad_splits(dt, n_out, n_in, nk);
    TC_REQUIRE((size_t)splits * n_out * n_in <= w.slab_floats, "linear_wgrad: slab buffer too small");
    if (dt == TC_BF16) {
        // row-major operands straight into the transposing-read GEMM
        const void *dyb = dy;
        if (in_dt == TC_F32) {
            TC_REQUIRE(dy_rowmajor != nullptr, "linear_wgrad: f32 gradient needs a bf16 staging buffer");
            TC_TRY(tc_colsum(TC_F32, dy, ld_dy, dy_rowmajor, n_out, dbias, M, n_out, stream));
            dyb = dy_rowmajor;
            ld_dy = n_out;
        } else if (dbias) {      // null: the producer of dY already added its column sums to the bias gradient
            TC_TRY(tc_colsum(TC_BF16, dy, ld_dy, nullptr, 0, dbias, M, n_out, stream));
        }
        TC_TRY(tc_gemm_tn(dyb, ld_dy, x, ld_x, w.slabs, n_in, M, n_out, n_in, splits, stream));
    } else {
        TC_TRY(tc_transpose_cast(in_dt, dt, dy, ld_dy, w.tr_a, Mp, dy_rowmajor, n_out, dbias, M, n_out, stream));
        TC_TRY(tc_transpose_cast(dt, dt, x, ld_x, w.tr_b, Mp, nullptr, 0, nullptr, M, n_in, stream));
        TC_TRY(tc_gemm_train(dt, TC_EPI_PLAIN_F32, w.tr_a, Mp, w.tr_b, nullptr, nullptr, w.slabs, nullptr, n_in,
                             n_out, n_in, Mp, splits, stream));
    }
    TC_TRY(tc_reduce_slabs(w.slabs, dW, splits, (long)n_out * n_in, 1, stream));
    return TC_OK;
}

// Transposed weights, wt[n_in, n_out] = W[n_out, n_in]^T: the "weight" operand of the data-gradient products dX = dY W.
// Layer l keeps q/k/v, out-proj, fc1, fc2 in this order; the two head dense matrices follow the last layer.
enum { WT_QKV = 0, WT_O = 1, WT_FC1 = 2, WT_FC2 = 3 };
void *wt_of(const tc_encoder_desc *d, const TrainWs &w, int layer, int which) {
    const size_t E = d->E, F = d->F, esz = d->dtype == TC_BF16 ? 2 : 4;
    const size_t per_layer = 4 * E * E + 2 * E * F;
    const size_t in_layer[4] = {0, 3 * E * E, 4 * E * E, 4 * E * E + E * F};
    return w.wt_all + ((size_t)layer * per_layer + in_layer[which]) * esz;
}
void *wt_of_head(const tc_encoder_desc *d, const TrainWs &w, int head) {
    const size_t E = d->E, F = d->F, esz = d->dtype == TC_BF16 ? 2 : 4;
    return w.wt_all + ((size_t)d->num_layers * (4 * E * E + 2 * E * F) + (size_t)head * E * E) * esz;
}
// one launch (per 64 matrices) for all of them
int transpose_weights(const tc_encoder_desc *d, const TrainWs &w, void *stream) {
    constexpr int kMost = 4 * kMaxLayers + 2;
    static_assert(kMost <= 512, "descriptor arrays on the stack");
    const void *src[kMost];
    void *dst[kMost];
    int rows[kMost], cols[kMost];
    const int E = d->E, F = d->F;
    const size_t esz = d->dtype == TC_BF16 ? 2 : 4;
    int n = 0;
    auto add = [&](const void *W, void *wt, int n_out, int n_in) { src[n] = W; dst[n] = wt; rows[n] = n_out; cols[n] = n_in; ++n; };
    for (int h = 0; h < 2; ++h)       // the backward pass starts with the heads
        add(static_cast<const char *>(d->w_heads) + (size_t)h * E * E * esz, wt_of_head(d, w, h), E, E);
    for (int l = d->num_layers - 1; l >= 0; --l) {
        const tc_layer_weights &lw = d->layers[l];
        add(lw.w_fc2, wt_of(d, w, l, WT_FC2), E, F);
        add(lw.w_fc1, wt_of(d, w, l, WT_FC1), F, E);
        add(lw.w_o, wt_of(d, w, l, WT_O), E, E);
        add(lw.w_qkv, wt_of(d, w, l, WT_QKV), 3 * E, E);
    }
    return tc_transpose_batch(d->dtype, n, src, dst, rows, cols, stream);
}

}  // namespace

extern "C" size_t tc_train_workspace_bytes(const tc_encoder_desc *desc, int B, int T) {
    if (!desc || B <= 0 || T <= 0 || desc->num_layers > kMaxLayers) return 0;
    return carve(desc, B, T, nullptr).total;
}

extern "C" int tc_encoder_forward_train(const tc_encoder_desc *d, const int32_t *tokens, const int32_t *species,
                                        int B, int T, float *logits, float *struct_logits, int32_t *status,
                                        void *workspace, size_t workspace_bytes, void *stream) {
    TC_TRY(check_desc(d, "tc_encoder_forward_train"));
    TC_REQUIRE(tokens && species && logits && struct_logits && workspace, "tc_encoder_forward_train: null operand");
    TC_REQUIRE(B > 0 && T > 0 && (long)B * T < (1L << 31) - 64, "tc_encoder_forward_train: bad batch B=%d T=%d", B, T);
    TC_REQUIRE(T <= d->rope_table_len, "tc_encoder_forward_train: T=%d exceeds the rope table (%d rows)", T,
               d->rope_table_len);
    TC_REQUIRE(tc_aligned16(workspace), "tc_encoder_forward_train: workspace must be 16-byte aligned");
    TrainWs w = carve(d, B, T, static_cast<char *>(workspace));
    if (w.total > workspace_bytes) {
        tc_set_error("tc_encoder_forward_train: workspace %zu bytes, need %zu", workspace_bytes, w.total);
        return TC_ENOSPC;
    }
    const int M = B * T, E = d->E, F = d->F, dt = d->dtype;
    const bool bf = dt == TC_BF16;
    const size_t esz = bf ? 2 : 4;
    auto ln = [&](const float *x, const float *g, const float *b, void *out) {
        return tc_layernorm_fwd(x, g, b, bf ? nullptr : static_cast<float *>(out), bf ? out : nullptr, M, E, d->ln_eps,
                                stream);
    };

    // The residual stream is updated out of place: each layer reads x_in, writes x_mid, and
    // writes the next layer's x_in (or w.x after the last layer).
    TC_TRY(tc_embed_fwd(tokens, species, d->tok_table, d->species_table, w.L[0].x_in, w.pad, status, B, T, E,
                        d->vocab, d->n_species, d->pad_idx, stream));
    TC_TRY(tc_key_mask_words(w.pad, w.words, B, T, stream));
    for (int l = 0; l < d->num_layers; ++l) {
        const tc_layer_weights &lw = d->layers[l];
        LayerSaved &sv = w.L[l];
        float *x_next = l + 1 < d->num_layers ? w.L[l + 1].x_in : w.x;
        TC_REQUIRE(lw.ln1_g && lw.ln1_b && lw.w_qkv && lw.b_qkv && lw.w_o && lw.b_o && lw.ln2_g && lw.ln2_b &&
                       lw.w_fc1 && lw.b_fc1 && lw.w_fc2 && lw.b_fc2 && lw.rope_cos && lw.rope_sin,
                   "tc_encoder_forward_train: layer %d has a null weight pointer", l);
        TC_TRY(ln(sv.x_in, lw.ln1_g, lw.ln1_b, sv.h1));
        TC_TRY(tc_gemm(dt, TC_EPI_QKV_ROPE, sv.h1, E, lw.w_qkv, lw.b_qkv, nullptr, sv.qkv, 3 * E, M, 3 * E, E, 0.125f,
                       lw.rope_cos, lw.rope_sin, T, 64, stream));
        TC_TRY(tc_attn_fwd_train(dt, sv.qkv, 3 * E, w.words, sv.ao, sv.lse, B, T, d->H, 64, stream));
        TC_TRY(tc_gemm(dt, TC_EPI_BIAS_RESIDUAL, sv.ao, E, lw.w_o, lw.b_o, sv.x_in, sv.x_mid, E, M, E, E, 0.f, nullptr,
                       nullptr, 0, 0, stream));
        TC_TRY(ln(sv.x_mid, lw.ln2_g, lw.ln2_b, sv.h2));
        TC_TRY(tc_gemm_train(dt, TC_EPI_BIAS_GELU_SAVE, sv.h2, E, lw.w_fc1, lw.b_fc1, nullptr, sv.mid, sv.u, w.ld_f, M, F,
                             E, 1, stream));
        TC_TRY(tc_gemm(dt, TC_EPI_BIAS_RESIDUAL, sv.mid, w.ld_f, lw.w_fc2, lw.b_fc2, sv.x_mid, x_next, E, M, E, F, 0.f,
                       nullptr, nullptr, 0, 0, stream));
    }
    TC_TRY(ln(w.x, d->lnf_g, d->lnf_b, w.hf));
    const char *w_heads = static_cast<const char *>(d->w_heads);
    TC_TRY(tc_gemm_train(dt, TC_EPI_BIAS_GELU_SAVE, w.hf, E, w_heads, d->b_heads, nullptr, w.g_lm, w.u_lm, E, M, E, E,
                         1, stream));
    TC_TRY(tc_gemm_train(dt, TC_EPI_BIAS_GELU_SAVE, w.hf, E, w_heads + (size_t)E * E * esz, d->b_heads + E, nullptr,
                         w.g_st, w.u_st, E, M, E, E, 1, stream));
    TC_TRY(tc_head_logits_fwd(dt, w.g_lm, E, 0, d->lm_ln_g, d->lm_ln_b, d->tok_table, d->lm_bias, logits, M, E,
                              d->vocab, d->ln_eps, stream));
    TC_TRY(tc_head_logits_fwd(dt, w.g_st, E, 0, d->st_ln_g, d->st_ln_b, d->st_w, d->st_bias, struct_logits, M, E,
                              d->n_struct, d->ln_eps, stream));
    return TC_OK;
}

extern "C" int tc_encoder_backward(const tc_encoder_desc *d, const tc_encoder_grads *g, const int32_t *tokens,
                                   const int32_t *species, int B, int T, const float *dlogits,
                                   const float *dstruct_logits, void *workspace, size_t workspace_bytes,
                                   void *stream) {
    return tc_encoder_backward_notify(d, g, tokens, species, B, T, dlogits, dstruct_logits, workspace, workspace_bytes,
                                      nullptr, nullptr, stream);
}

extern "C" int tc_encoder_backward_notify(const tc_encoder_desc *d, const tc_encoder_grads *g, const int32_t *tokens,
                                          const int32_t *species, int B, int T, const float *dlogits,
                                          const float *dstruct_logits, void *workspace, size_t workspace_bytes,
                                          tc_grads_ready_fn ready, void *user, void *stream) {
    TC_TRY(check_desc(d, "tc_encoder_backward"));
    TC_REQUIRE(g && g->layers && tokens && species && dlogits && workspace, "tc_encoder_backward: null operand");
    TC_REQUIRE(B > 0 && T > 0 && (long)B * T < (1L << 31) - 64, "tc_encoder_backward: bad batch B=%d T=%d", B, T);
    TC_REQUIRE(g->tok_table && g->species_table && g->lnf_g && g->lnf_b && g->w_heads && g->b_heads && g->lm_ln_g &&
                   g->lm_ln_b && g->lm_bias && g->st_ln_g && g->st_ln_b && g->st_w && g->st_bias,
               "tc_encoder_backward: null gradient pointer");
    TC_REQUIRE(T <= d->rope_table_len, "tc_encoder_backward: T=%d exceeds the rope table (%d rows)", T,
               d->rope_table_len);
    TC_REQUIRE(tc_aligned16(workspace), "tc_encoder_backward: workspace must be 16-byte aligned");
    // every pointer is checked before the first launch: a failure below can then only be a launch error
    for (int l = 0; l < d->num_layers; ++l) {
        const tc_layer_weights &lw = d->layers[l];
        const tc_layer_grads &lg = g->layers[l];
        TC_REQUIRE(lw.ln1_g && lw.ln1_b && lw.w_qkv && lw.b_qkv && lw.w_o && lw.b_o && lw.ln2_g && lw.ln2_b &&
                       lw.w_fc1 && lw.b_fc1 && lw.w_fc2 && lw.b_fc2 && lw.rope_cos && lw.rope_sin,
                   "tc_encoder_backward: layer %d has a null weight pointer", l);
        TC_REQUIRE(lg.ln1_g && lg.ln1_b && lg.w_qkv && lg.b_qkv && lg.w_o && lg.b_o && lg.ln2_g && lg.ln2_b &&
                       lg.w_fc1 && lg.b_fc1 && lg.w_fc2 && lg.b_fc2,
                   "tc_encoder_backward: layer %d has a null gradient pointer", l);
    }
    TrainWs w = carve(d, B, T, static_cast<char *>(workspace));
    if (w.total > workspace_bytes) {
        tc_set_error("tc_encoder_backward: workspace %zu bytes, need %zu", workspace_bytes, w.total);
        return TC_ENOSPC;
    }
    const int M = B * T, E = d->E, F = d->F, dt = d->dtype;
    float *dx = w.t1;    // gradient of the residual stream, walked down the layers

    // ---- heads: d_hf accumulated in t0 ----
    struct Head {
        const float *dl; const void *g_act; const void *u; const float *ln_g, *ln_b, *W;
        float *d_ln_g, *d_ln_b, *dW, *dbias; int V; size_t wofs;
    };
    const Head heads[2] = {
        {dlogits, w.g_lm, w.u_lm, d->lm_ln_g, d->lm_ln_b, d->tok_table, g->lm_ln_g, g->lm_ln_b, g->tok_table,
         g->lm_bias, d->vocab, 0},
        {dstruct_logits, w.g_st, w.u_st, d->st_ln_g, d->st_ln_b, d->st_w, g->st_ln_g, g->st_ln_b, g->st_w, g->st_bias,
         d->n_struct, (size_t)E * E},
    };
    // every data-gradient product below reads its weight matrix transposed
    TC_TRY(transpose_weights(d, w, stream));
    bool have_dhf = false;
    for (const Head &h : heads) {
        if (!h.dl) continue;
        // logits -> LayerNorm output (wide as f32 scratch) -> gelu output (t1) -> pre-activation (dyb)
        float *d_ln = reinterpret_cast<float *>(w.wide);
        TC_TRY(tc_head_logits_bwd(dt, h.g_act, E, 0, h.ln_g, h.ln_b, h.W, h.dl, d_ln, h.dW, h.dbias, M, E, h.V,
                                  d->ln_eps, stream));
        TC_TRY(tc_layernorm_bwd(dt, h.g_act, E, h.ln_g, d_ln, w.t1, 0, h.d_ln_g, h.d_ln_b, M, E, d->ln_eps, stream));
        TC_TRY(tc_dgelu(dt, w.t1, h.u, w.dyb, (long)M * E, stream));
        TC_TRY(linear_wgrad(d, w, dt, w.dyb, E, nullptr, w.hf, E, M, E, E, g->w_heads + h.wofs, g->b_heads + (h.wofs ? E : 0),
                            stream));
        const void *wt = wt_of_head(d, w, h.wofs ? 1 : 0);
        if (!have_dhf)
            TC_TRY(tc_gemm_train(dt, TC_EPI_PLAIN_F32, w.dyb, E, wt, nullptr, nullptr, w.t0, nullptr, E, M, E, E, 1,
                                 stream));
        else
            TC_TRY(tc_gemm(dt, TC_EPI_BIAS_RESIDUAL, w.dyb, E, wt, nullptr, w.t0, w.t0, E, M, E, E, 0.f, nullptr,
                           nullptr, 0, 0, stream));
        have_dhf = true;
    }
    TC_REQUIRE(have_dhf, "tc_encoder_backward: no output gradient");
    // final LayerNorm (input: the forward's last residual state, still in w.x)
    // bf16 mode: every LayerNorm backward also leaves the bf16 copy of the residual-stream gradient it
    // has just completed (w.dyb) and adds its column sums to the bias gradient of the Linear layer below
    // (fc2 of the layer underneath, or this layer's out-proj), which saves a pass over dx per Linear.
    const bool fuse = dt == TC_BF16;
    TC_TRY(tc_layernorm_bwd_dy(TC_F32, w.x, E, d->lnf_g, TC_F32, w.t0, E, dx, 0, g->lnf_g, g->lnf_b,
                               fuse ? w.dyb : nullptr, fuse ? g->layers[d->num_layers - 1].b_fc2 : nullptr, M, E,
                               d->ln_eps, stream));

    for (int l = d->num_layers - 1; l >= 0; --l) {
        const tc_layer_weights &lw = d->layers[l];
        const tc_layer_grads &lg = g->layers[l];
        LayerSaved &sv = w.L[l];
        // ---- feed-forward block: x_out = x_mid + fc2(gelu(fc1(LN2(x_mid)))) ----
        if (fuse)
            TC_TRY(linear_wgrad(d, w, TC_BF16, w.dyb, E, nullptr, sv.mid, w.ld_f, M, E, F, lg.w_fc2, nullptr, stream));
        else
            TC_TRY(linear_wgrad(d, w, TC_F32, dx, E, w.dyb, sv.mid, w.ld_f, M, E, F, lg.w_fc2, lg.b_fc2, stream));
        // fc1's bias gradient is summed in the epilogue that produces d_u (bf16 mode)
        TC_TRY(tc_gemm_train(dt, TC_EPI_DGELU, w.dyb, E, wt_of(d, w, l, WT_FC2), nullptr, sv.u, w.wide, fuse ? lg.b_fc1 : nullptr, w.ld_f, M,
                             F, E, 1, stream));
        TC_TRY(linear_wgrad(d, w, dt, w.wide, w.ld_f, nullptr, sv.h2, E, M, F, E, lg.w_fc1, fuse ? nullptr : lg.b_fc1,
                            stream));
        // the gradient handed to a LayerNorm comes out of the product in the compute type (bf16: what a
        // bf16 Linear returns under autocast), half the bytes for the product to write and the LayerNorm to read
        TC_TRY(tc_gemm(dt, TC_EPI_BIAS, w.wide, w.ld_f, wt_of(d, w, l, WT_FC1), nullptr, nullptr, w.t0, E, M, E, F, 0.f, nullptr, nullptr, 0,
                       0, stream));
        TC_TRY(tc_layernorm_bwd_dy(TC_F32, sv.x_mid, E, lw.ln2_g, dt, w.t0, E, dx, 1, lg.ln2_g, lg.ln2_b,
                                   fuse ? w.dyb : nullptr, fuse ? lg.b_o : nullptr, M, E, d->ln_eps, stream));
        // ---- attention block: x_mid = x_in + out_proj(attn(rope(qkv(LN1(x_in))))) ----
        if (fuse)
            TC_TRY(linear_wgrad(d, w, TC_BF16, w.dyb, E, nullptr, sv.ao, E, M, E, E, lg.w_o, nullptr, stream));
        else
            TC_TRY(linear_wgrad(d, w, TC_F32, dx, E, w.dyb, sv.ao, E, M, E, E, lg.w_o, lg.b_o, stream));
        void *d_ao = w.tr_b;      // free between weight-gradient products
        TC_TRY(tc_gemm(dt, TC_EPI_BIAS, w.dyb, E, wt_of(d, w, l, WT_O), nullptr, nullptr, d_ao, E, M, E, E, 0.f, nullptr, nullptr, 0, 0,
                       stream));
        TC_TRY(tc_attn_bwd(dt, sv.qkv, 3 * E, w.words, sv.ao, d_ao, sv.lse, w.delta, w.wide, 3 * E, lw.rope_cos,
                           lw.rope_sin, 0.125f, B, T, d->H, 64, stream));
        TC_TRY(linear_wgrad(d, w, dt, w.wide, 3 * E, nullptr, sv.h1, E, M, 3 * E, E, lg.w_qkv, lg.b_qkv, stream));
        TC_TRY(tc_gemm(dt, TC_EPI_BIAS, w.wide, 3 * E, wt_of(d, w, l, WT_QKV), nullptr, nullptr, w.t0, E, M, E, 3 * E, 0.f, nullptr, nullptr,
                       0, 0, stream));
        const bool below = fuse && l > 0;    // layer 0 hands dx to the embedding backward only
        TC_TRY(tc_layernorm_bwd_dy(TC_F32, sv.x_in, E, lw.ln1_g, dt, w.t0, E, dx, 1, lg.ln1_g, lg.ln1_b,
                                   below ? w.dyb : nullptr, below ? g->layers[l - 1].b_fc2 : nullptr, M, E,
                                   d->ln_eps, stream));
        if (ready) ready(l, user);       // every kernel that writes layer l's gradients is enqueued
    }
    TC_TRY(tc_embed_bwd(tokens, species, dx, g->tok_table, g->species_table, B, T, E, d->vocab, d->n_species,
                        d->pad_idx, stream));
    if (ready) ready(-1, user);          // embeddings, final LayerNorm and heads
    return TC_OK;
}
