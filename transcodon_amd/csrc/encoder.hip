// Whole-encoder forward: strings the kernels together on one stream, no host sync.
#include "common.h"

namespace {

inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

struct Workspace {
    float *x;          // residual stream f32 [M,E]
    void *h;           // LayerNorm output / attention output [M,E] (dtype)
    void *qkv;         // [M,3E] (dtype)
    void *mid;         // [M,max(F,2E)] (dtype), rows ld_mid elements apart
    int ld_mid;
    uint8_t *pad;      // [M]
    uint64_t *words;   // [B, ceil(T/64)]
    // LayerNorms folded into the products (bf16 only): bf16 copy of the residual stream, per row
    // (rstd, -rstd * mean) and the integer accumulators they are made from, rows rounded up to 256
    void *xb;
    float *ln_rows;
    unsigned long long *row_acc;
    size_t rows_pad;
    size_t total;
};

Workspace carve(const tc_encoder_desc *d, int B, int T, char *base) {
    const size_t M = (size_t)B * T;
    const size_t esz = d->dtype == TC_BF16 ? 2 : 4;
    const size_t wide = (size_t)(d->F > 2 * d->E ? d->F : 2 * d->E);
    Workspace w{};
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char *p = base ? base + off : nullptr;
        off += align_up(bytes);
        return p;
    };
    w.x = reinterpret_cast<float *>(take(M * d->E * 4));
    w.h = take(M * d->E * esz);
    w.qkv = take(M * 3 * d->E * esz);
    // Rows of the FFN intermediate are padded by 128 bytes: with a row stride of 6144 bytes (F =
    // 3072 in bf16) the stores of the fc1 epilogue run at a fraction of their normal rate on
    // this memory system (measured: 672 us against 579 us for the 131072 x 3072 x 768 GEMM).
    w.ld_mid = (int)wide + 64;
    w.mid = take(M * (size_t)w.ld_mid * esz);
    w.pad = reinterpret_cast<uint8_t *>(take(M));
    w.words = reinterpret_cast<uint64_t *>(take((size_t)B * ((T + 63) / 64) * 8));
    if (d->dtype == TC_BF16) {
        w.rows_pad = (M + 255) / 256 * 256;
        w.xb = take(M * d->E * 2);
        w.ln_rows = reinterpret_cast<float *>(take(w.rows_pad * 8));
        w.row_acc = reinterpret_cast<unsigned long long *>(take(w.rows_pad * 16));
    }
    w.total = off;
    return w;
}

int check_desc(const tc_encoder_desc *d) {
    TC_REQUIRE(d != nullptr, "encoder: null descriptor");
    TC_REQUIRE(d->dtype == TC_BF16 || d->dtype == TC_F32, "encoder: bad dtype %d", d->dtype);
    TC_REQUIRE(d->num_layers > 0 && d->layers, "encoder: no layers");
    TC_REQUIRE(d->E > 0 && d->H > 0 && d->E == d->H * 64, "encoder: E=%d H=%d needs head_dim 64", d->E, d->H);
    TC_REQUIRE(d->E % 128 == 0 && d->F % 128 == 0, "encoder: E=%d and F=%d must be multiples of 128", d->E, d->F);
    TC_REQUIRE(d->vocab > 0 && d->vocab <= 16 && d->n_struct >= 0 && d->n_struct <= 16, "encoder: head widths out of range");
    TC_REQUIRE(d->tok_table && d->species_table && d->lnf_g && d->lnf_b && d->w_heads && d->b_heads &&
                   d->lm_ln_g && d->lm_ln_b && d->lm_bias,
               "encoder: null weight pointer");
    return TC_OK;
}

}  // namespace

extern "C" size_t tc_workspace_bytes(const tc_encoder_desc *desc, int B, int T) {
    if (!desc || B <= 0 || T <= 0) return 0;
    return carve(desc, B, T, nullptr).total;
}

#define TC_TRY(expr)                 \
    do {                             \
        int rc__ = (expr);           \
        if (rc__ != TC_OK) return rc__; \
    } while (0)

// x += A W^T + b on the f32 residual stream (modules.py:96-97,102-104).  A short f32 batch has only a few dozen
// 128 x 128 tiles of N = E columns: its k range is then cut in three, the parts land in the q/k/v buffer (free
// once attention has run: 3 x [M, E] is exactly its size) and one small kernel adds them, the bias and x.
static int residual_product_f32x(int dt, const Workspace &w, const void *A, int lda, const void *W, const float *bias, int M,
                                 int E, int K, void *stream) {
    if (dt != TC_F32 || M > 2048)
        return tc_gemm(dt, TC_EPI_BIAS_RESIDUAL, A, lda, W, bias, w.x, w.x, E, M, E, K, 0.f, nullptr, nullptr, 0, 0, stream);
    TC_TRY(tc_gemm_train(dt, TC_EPI_PLAIN_F32, A, lda, W, nullptr, nullptr, w.qkv, nullptr, E, M, E, K, 3, stream));
    return tc_splitk_bias_residual(static_cast<const float *>(w.qkv), 3, bias, w.x, M, E, stream);
}

extern "C" int tc_encoder_forward(const tc_encoder_desc *d, const int32_t *tokens,
                                  const int32_t *species, int B, int T, float *logits,
                                  float *struct_logits, float *rep_last, int n_rep,
                                  const int32_t *rep_layer, float *const *rep_out, int32_t *status,
                                  void *workspace, size_t workspace_bytes, void *stream) {
    TC_TRY(check_desc(d));
    TC_REQUIRE(tokens && species && logits && workspace, "tc_encoder_forward: null operand");
    TC_REQUIRE(B > 0 && T > 0 && (long)B * T < (1L << 31), "tc_encoder_forward: bad batch B=%d T=%d", B, T);
    TC_REQUIRE(T <= d->rope_table_len, "tc_encoder_forward: T=%d exceeds the rope table (%d rows)", T, d->rope_table_len);
    TC_REQUIRE(!struct_logits || (d->st_ln_g && d->st_ln_b && d->st_w && d->st_bias && d->n_struct > 0),
               "tc_encoder_forward: structure head requested without its weights");
    TC_REQUIRE(tc_aligned16(workspace), "tc_encoder_forward: workspace must be 16-byte aligned");
    Workspace w = carve(d, B, T, static_cast<char *>(workspace));
    if (w.total > workspace_bytes) {
        tc_set_error("tc_encoder_forward: workspace %zu bytes, need %zu", workspace_bytes, w.total);
        return TC_ENOSPC;
    }
    for (int i = 0; i < n_rep; ++i)
        TC_REQUIRE(rep_layer && rep_out && rep_layer[i] >= 0 && rep_layer[i] < d->num_layers && rep_out[i],
                   "tc_encoder_forward: bad representation request %d", i);

    const int M = B * T, E = d->E, F = d->F, dt = d->dtype;
    const bool bf = dt == TC_BF16;
    // bf16: q leaves the q/k/v product scaled by log2(e) / sqrt(64), the base-2 form the attention kernel takes
    const float q_scale = bf ? 0.125f * 1.4426950408889634f : 0.125f;
    auto attn = [&]() {
        return bf ? tc_attn_fwd_q2(dt, w.qkv, 3 * E, w.words, w.h, B, T, d->H, 64, stream)
                  : tc_attn_fwd(dt, w.qkv, 3 * E, w.words, w.h, B, T, d->H, 64, stream);
    };
    auto ln = [&](const float *g, const float *b, float *extra_f32) {
        return tc_layernorm_fwd(w.x, g, b, bf ? extra_f32 : static_cast<float *>(w.h),
                                bf ? w.h : nullptr, M, E, d->ln_eps, stream);
    };
    auto residual_product = [&](const void *A, int lda, const void *W, const float *bias, int K) {
        return residual_product_f32x(dt, w, A, lda, W, bias, M, E, K, stream);
    };
    void *x_lo = w.x;          // folded forward: the residual stream is the plane pair (w.xb, x_lo), no f32 copy exists
    bool fold = false;
    auto save_rep = [&](int layer) -> int {
        for (int i = 0; i < n_rep; ++i)
            if (rep_layer[i] == layer) {
                if (fold) {
                    TC_TRY(tc_split_to_f32(w.xb, x_lo, rep_out[i], (long)M * E, stream));
                    continue;
                }
                TC_TRY(tc_copy_f32(rep_out[i], w.x, (long)M * E, stream));
            }
        return TC_OK;
    };

    // Folded LayerNorms: when every layer carries the folded weights and the batch is large enough
    // for the alternating large-tile kernels, no LayerNorm kernel runs inside the layers.
    fold = bf && tc_ln_fold_supported(dt, B, T, E, F) != 0;
    for (int l = 0; l < d->num_layers && fold; ++l) {
        const tc_layer_weights &lw = d->layers[l];
        fold = lw.w_qkv_ln && lw.c_qkv_ln && lw.s_qkv_ln && lw.w_fc1_ln && lw.c_fc1_ln && lw.s_fc1_ln;
    }
    // emb_layer_norm_after folded into the head dense layers as well: nobody asked for its output
    const bool fold_heads = fold && !rep_last && d->w_heads_ln && d->c_heads_ln && d->s_heads_ln;
    if (fold) {
        // (a kernel: a memset node of a captured graph is not ordered against the graph's previous launch)
        TC_TRY(tc_ln_rows_clear(w.row_acc, (int)w.rows_pad, stream));
        TC_TRY(tc_embed_fwd_ln(tokens, species, d->tok_table, d->species_table, nullptr, w.pad, status, B, T, E, d->vocab,
                               d->n_species, d->pad_idx, w.xb, x_lo, w.ln_rows, d->ln_eps, stream));
    } else {
        TC_TRY(tc_embed_fwd(tokens, species, d->tok_table, d->species_table, w.x, w.pad, status, B, T, E,
                            d->vocab, d->n_species, d->pad_idx, stream));
    }
    TC_TRY(tc_key_mask_words(w.pad, w.words, B, T, stream));
    TC_TRY(save_rep(0));

    for (int l = 0; l < d->num_layers; ++l) {
        const tc_layer_weights &lw = d->layers[l];
        TC_REQUIRE(lw.ln1_g && lw.ln1_b && lw.w_qkv && lw.b_qkv && lw.w_o && lw.b_o && lw.ln2_g && lw.ln2_b &&
                       lw.w_fc1 && lw.b_fc1 && lw.w_fc2 && lw.b_fc2 && lw.rope_cos && lw.rope_sin,
                   "tc_encoder_forward: layer %d has a null weight pointer", l);
        if (fold) {
            // x, bf16(x) and the row statistics of bf16(x) travel together; LN1 and LN2 are applied
            // inside the q/k/v and fc1 products (modules.py:87,100)
            const bool last = l + 1 == d->num_layers;
            TC_TRY(tc_gemm_ln(dt, TC_EPI_QKV_ROPE, w.xb, E, lw.w_qkv_ln, lw.c_qkv_ln, lw.s_qkv_ln, w.ln_rows, w.qkv,
                              3 * E, M, 3 * E, E, q_scale, lw.rope_cos, lw.rope_sin, T, 64, stream));
            TC_TRY(attn());
            TC_TRY(tc_gemm_residual_split(dt, w.h, E, lw.w_o, lw.b_o, w.xb, x_lo, w.row_acc, E, M, E, E, stream));
            TC_TRY(tc_ln_rows_finalize(w.row_acc, w.ln_rows, M, E, d->ln_eps, stream));
            TC_TRY(tc_gemm_ln(dt, TC_EPI_BIAS_GELU, w.xb, E, lw.w_fc1_ln, lw.c_fc1_ln, lw.s_fc1_ln, w.ln_rows, w.mid,
                              w.ld_mid, M, F, E, 0.f, nullptr, nullptr, 0, 0, stream));
            TC_TRY(tc_gemm_residual_split(dt, w.mid, w.ld_mid, lw.w_fc2, lw.b_fc2, w.xb, x_lo, w.row_acc, E, M, E, F,
                                          stream));
            // (after the last layer the row sums serve the head product when the final LayerNorm is folded into it)
            if (!last || fold_heads) TC_TRY(tc_ln_rows_finalize(w.row_acc, w.ln_rows, M, E, d->ln_eps, stream));
            if (!last) TC_TRY(save_rep(l + 1));
            continue;
        }
        TC_TRY(ln(lw.ln1_g, lw.ln1_b, nullptr));
        TC_TRY(tc_gemm(dt, TC_EPI_QKV_ROPE, w.h, E, lw.w_qkv, lw.b_qkv, nullptr, w.qkv, 3 * E, M, 3 * E, E,
                       q_scale, lw.rope_cos, lw.rope_sin, T, 64, stream));
        TC_TRY(attn());
        TC_TRY(residual_product(w.h, E, lw.w_o, lw.b_o, E));
        TC_TRY(ln(lw.ln2_g, lw.ln2_b, nullptr));
        TC_TRY(tc_gemm(dt, TC_EPI_BIAS_GELU, w.h, E, lw.w_fc1, lw.b_fc1, nullptr, w.mid, w.ld_mid, M, F, E, 0.f,
                       nullptr, nullptr, 0, 0, stream));
        TC_TRY(residual_product(w.mid, w.ld_mid, lw.w_fc2, lw.b_fc2, F));
        if (l + 1 < d->num_layers) TC_TRY(save_rep(l + 1));
    }

    // final LayerNorm; the f32 copy doubles as representations[num_layers]
    const int nheads = struct_logits ? 2 : 1;
    if (fold_heads) {
        TC_TRY(tc_gemm_ln(dt, TC_EPI_BIAS_GELU, w.xb, E, d->w_heads_ln, d->c_heads_ln, d->s_heads_ln, w.ln_rows, w.mid,
                          w.ld_mid, M, nheads * E, E, 0.f, nullptr, nullptr, 0, 0, stream));
    } else if (fold) {
        TC_TRY(tc_layernorm_fwd_split(w.xb, x_lo, d->lnf_g, d->lnf_b, rep_last, w.h, M, E, d->ln_eps, stream));
    } else if (bf) {
        TC_TRY(tc_layernorm_fwd(w.x, d->lnf_g, d->lnf_b, rep_last, w.h, M, E, d->ln_eps, stream));
    } else {
        TC_TRY(tc_layernorm_fwd(w.x, d->lnf_g, d->lnf_b, static_cast<float *>(w.h), nullptr, M, E, d->ln_eps, stream));
        if (rep_last) {
            TC_TRY(tc_copy_f32(rep_last, static_cast<const float *>(w.h), (long)M * E, stream));
        }
    }
    if (!fold_heads)
        TC_TRY(tc_gemm(dt, TC_EPI_BIAS_GELU, w.h, E, d->w_heads, d->b_heads, nullptr, w.mid, w.ld_mid, M,
                       nheads * E, E, 0.f, nullptr, nullptr, 0, 0, stream));
    TC_TRY(tc_head_logits_fwd(dt, w.mid, w.ld_mid, 0, d->lm_ln_g, d->lm_ln_b, d->tok_table, d->lm_bias,
                              logits, M, E, d->vocab, d->ln_eps, stream));
    if (struct_logits)
        TC_TRY(tc_head_logits_fwd(dt, w.mid, w.ld_mid, E, d->st_ln_g, d->st_ln_b, d->st_w, d->st_bias,
                                  struct_logits, M, E, d->n_struct, d->ln_eps, stream));
    return TC_OK;
}

// Packed (variable-length) forward: the sequences lie one after the other in the row dimension, no
// padding row exists, so every row kernel and GEMM works on sum(T_s) rows.  Rotary positions restart
// per sequence (pos), attention runs per (sequence, query block) from the block list.
extern "C" int tc_encoder_forward_packed(const tc_encoder_desc *d, const int32_t *tokens, const int32_t *species_row,
                                         const int32_t *pos, const int32_t *attn_blocks, int n_blocks,
                                         const int32_t *row_panels, int n_panels,
                                         const int32_t *cu_seqlens_host, int n_seq, float *logits,
                                         float *struct_logits, float *rep_last, int32_t *status, void *workspace,
                                         size_t workspace_bytes, void *stream) {
    TC_TRY(check_desc(d));
    TC_REQUIRE(tokens && species_row && pos && attn_blocks && cu_seqlens_host && workspace,
               "tc_encoder_forward_packed: null operand");
    TC_REQUIRE(logits || rep_last, "tc_encoder_forward_packed: no output requested");
    TC_REQUIRE(n_seq > 0 && cu_seqlens_host[0] == 0, "tc_encoder_forward_packed: cu_seqlens must start at 0");
    const int qrows = tc_attn_packed_block_rows(d->dtype);
    long want_blocks = 0, want_panels = 0;
    int max_len = 0;
    double attn_flops = 0.0;
    for (int i = 0; i < n_seq; ++i) {
        const long len = (long)cu_seqlens_host[i + 1] - cu_seqlens_host[i];
        TC_REQUIRE(len > 0 && len <= d->rope_table_len,
                   "tc_encoder_forward_packed: sequence %d has %ld tokens (1..%d, the rope table length)", i, len,
                   d->rope_table_len);
        want_blocks += (len + qrows - 1) / qrows;
        want_panels += (len + 255) / 256;
        if (len > max_len) max_len = (int)len;
        attn_flops += 4.0 * d->H * 64.0 * (double)len * (double)len;
    }
    TC_REQUIRE(want_blocks == n_blocks, "tc_encoder_forward_packed: block list has %d entries, the lengths need %ld",
               n_blocks, want_blocks);
    TC_REQUIRE(!row_panels || (want_panels == n_panels && tc_aligned16(row_panels)),
               "tc_encoder_forward_packed: panel list has %d entries, the lengths need %ld (16-byte aligned)", n_panels,
               want_panels);
    const long Ml = cu_seqlens_host[n_seq];
    TC_REQUIRE(Ml > 0 && Ml < (1L << 31), "tc_encoder_forward_packed: bad row count %ld", Ml);
    TC_REQUIRE(!struct_logits || (d->st_ln_g && d->st_ln_b && d->st_w && d->st_bias && d->n_struct > 0),
               "tc_encoder_forward_packed: structure head requested without its weights");
    TC_REQUIRE(tc_aligned16(workspace) && tc_aligned16(attn_blocks),
               "tc_encoder_forward_packed: workspace and block list must be 16-byte aligned");
    const int M = (int)Ml, E = d->E, F = d->F, dt = d->dtype;
    Workspace w = carve(d, 1, M, static_cast<char *>(workspace));
    if (w.total > workspace_bytes) {
        tc_set_error("tc_encoder_forward_packed: workspace %zu bytes, need %zu", workspace_bytes, w.total);
        return TC_ENOSPC;
    }
    const bool bf = dt == TC_BF16;
    auto ln = [&](const float *g, const float *b, float *extra_f32) {
        return tc_layernorm_fwd(w.x, g, b, bf ? extra_f32 : static_cast<float *>(w.h), bf ? w.h : nullptr, M, E,
                                d->ln_eps, stream);
    };
    for (int l = 0; l < d->num_layers; ++l) {
        const tc_layer_weights &lw = d->layers[l];
        TC_REQUIRE(lw.ln1_g && lw.ln1_b && lw.w_qkv && lw.b_qkv && lw.w_o && lw.b_o && lw.ln2_g && lw.ln2_b &&
                       lw.w_fc1 && lw.b_fc1 && lw.w_fc2 && lw.b_fc2 && lw.rope_cos && lw.rope_sin,
                   "tc_encoder_forward_packed: layer %d has a null weight pointer", l);
    }
    // Large bf16 batches whose layers carry the folded fc1 weights: the residual stream is the bf16 plane pair
    // (w.xb, w.x), the second LayerNorm of every layer is applied inside fc1 (modules.py:100-101) from the row
    // sums the out-proj epilogue leaves, and only the first LayerNorm stays a kernel (the packed q/k/v product
    // gathers its rotary factors per row and has no folded form).
    bool fold2 = bf && tc_ln_fold_rows_supported(dt, M, E, F) != 0;
    for (int l = 0; l < d->num_layers && fold2; ++l) {
        const tc_layer_weights &lw = d->layers[l];
        fold2 = lw.w_fc1_ln && lw.c_fc1_ln && lw.s_fc1_ln;
    }
    // With the list of row panels (every sequence cut into pieces of 256 rows) the q/k/v product also takes its
    // folded form, and no LayerNorm kernel runs inside the layers, as in the padded forward.
    bool fold1 = fold2 && row_panels != nullptr && d->rope_table_len >= 96;
    for (int l = 0; l < d->num_layers && fold1; ++l) {
        const tc_layer_weights &lw = d->layers[l];
        fold1 = lw.w_qkv_ln && lw.c_qkv_ln && lw.s_qkv_ln;
    }
    const float q_scale = bf ? 0.125f * 1.4426950408889634f : 0.125f;   // bf16: base-2 attention kernel
    auto attention = [&]() {
        return bf ? tc_attn_fwd_packed_q2(dt, w.qkv, 3 * E, attn_blocks, n_blocks, max_len, w.h, d->H, 64, attn_flops, stream)
                  : tc_attn_fwd_packed(dt, w.qkv, 3 * E, attn_blocks, n_blocks, max_len, w.h, d->H, 64, attn_flops, stream);
    };
    // one "sequence" per row for the gather: species_row[m] is the species of row m's sequence
    if (fold2) {
        void *x_lo = w.x;
        TC_TRY(tc_ln_rows_clear(w.row_acc, (int)w.rows_pad, stream));
        TC_TRY(tc_embed_fwd_ln(tokens, species_row, d->tok_table, d->species_table, nullptr, nullptr, status, M, 1, E,
                               d->vocab, d->n_species, d->pad_idx, w.xb, x_lo, w.ln_rows, d->ln_eps, stream));
        for (int l = 0; l < d->num_layers; ++l) {
            const tc_layer_weights &lw = d->layers[l];
            if (fold1) {
                TC_TRY(tc_gemm_ln_qkv_panels(dt, w.xb, E, lw.w_qkv_ln, lw.c_qkv_ln, lw.s_qkv_ln, w.ln_rows, w.qkv, 3 * E, M,
                                             3 * E, E, q_scale, lw.rope_cos, lw.rope_sin, d->rope_table_len, 64, row_panels,
                                             n_panels, stream));
            } else {
                TC_TRY(tc_layernorm_fwd_split(w.xb, x_lo, lw.ln1_g, lw.ln1_b, nullptr, w.h, M, E, d->ln_eps, stream));
                TC_TRY(tc_gemm_qkv_packed(dt, w.h, E, lw.w_qkv, lw.b_qkv, w.qkv, 3 * E, M, 3 * E, E, q_scale, lw.rope_cos,
                                          lw.rope_sin, pos, d->rope_table_len, 64, stream));
            }
            TC_TRY(attention());
            TC_TRY(tc_gemm_residual_split(dt, w.h, E, lw.w_o, lw.b_o, w.xb, x_lo, w.row_acc, E, M, E, E, stream));
            TC_TRY(tc_ln_rows_finalize(w.row_acc, w.ln_rows, M, E, d->ln_eps, stream));
            TC_TRY(tc_gemm_ln(dt, TC_EPI_BIAS_GELU, w.xb, E, lw.w_fc1_ln, lw.c_fc1_ln, lw.s_fc1_ln, w.ln_rows, w.mid,
                              w.ld_mid, M, F, E, 0.f, nullptr, nullptr, 0, 0, stream));
            // row sums only when the next layer's first LayerNorm is folded too (a LayerNorm kernel takes its own)
            const bool sums = fold1 && l + 1 < d->num_layers;
            TC_TRY(tc_gemm_residual_split(dt, w.mid, w.ld_mid, lw.w_fc2, lw.b_fc2, w.xb, x_lo, sums ? w.row_acc : nullptr, E, M,
                                          E, F, stream));
            if (sums) TC_TRY(tc_ln_rows_finalize(w.row_acc, w.ln_rows, M, E, d->ln_eps, stream));
        }
        TC_TRY(tc_layernorm_fwd_split(w.xb, x_lo, d->lnf_g, d->lnf_b, rep_last, (logits || !rep_last) ? w.h : nullptr, M, E,
                                      d->ln_eps, stream));
    } else {
        TC_TRY(tc_embed_fwd(tokens, species_row, d->tok_table, d->species_table, w.x, nullptr, status, M, 1, E, d->vocab,
                            d->n_species, d->pad_idx, stream));
        for (int l = 0; l < d->num_layers; ++l) {
            const tc_layer_weights &lw = d->layers[l];
            TC_TRY(ln(lw.ln1_g, lw.ln1_b, nullptr));
            TC_TRY(tc_gemm_qkv_packed(dt, w.h, E, lw.w_qkv, lw.b_qkv, w.qkv, 3 * E, M, 3 * E, E, q_scale, lw.rope_cos,
                                      lw.rope_sin, pos, d->rope_table_len, 64, stream));
            TC_TRY(attention());
            TC_TRY(residual_product_f32x(dt, w, w.h, E, lw.w_o, lw.b_o, M, E, E, stream));
            TC_TRY(ln(lw.ln2_g, lw.ln2_b, nullptr));
            TC_TRY(tc_gemm(dt, TC_EPI_BIAS_GELU, w.h, E, lw.w_fc1, lw.b_fc1, nullptr, w.mid, w.ld_mid, M, F, E, 0.f,
                           nullptr, nullptr, 0, 0, stream));
            TC_TRY(residual_product_f32x(dt, w, w.mid, w.ld_mid, lw.w_fc2, lw.b_fc2, M, E, F, stream));
        }
        if (bf) {
            TC_TRY(tc_layernorm_fwd(w.x, d->lnf_g, d->lnf_b, rep_last, (logits || !rep_last) ? w.h : nullptr, M, E,
                                    d->ln_eps, stream));
        } else {
            float *dst = logits ? static_cast<float *>(w.h) : rep_last;      // embedding extraction writes the result directly
            TC_TRY(tc_layernorm_fwd(w.x, d->lnf_g, d->lnf_b, dst, nullptr, M, E, d->ln_eps, stream));
            if (rep_last && dst != rep_last) {
                TC_TRY(tc_copy_f32(rep_last, static_cast<const float *>(w.h), (long)M * E, stream));
            }
        }
    }
    if (!logits) return TC_OK;              // embedding extraction needs no head
    const int nheads = struct_logits ? 2 : 1;
    TC_TRY(tc_gemm(dt, TC_EPI_BIAS_GELU, w.h, E, d->w_heads, d->b_heads, nullptr, w.mid, w.ld_mid, M, nheads * E, E, 0.f,
                   nullptr, nullptr, 0, 0, stream));
    TC_TRY(tc_head_logits_fwd(dt, w.mid, w.ld_mid, 0, d->lm_ln_g, d->lm_ln_b, d->tok_table, d->lm_bias, logits, M, E,
                              d->vocab, d->ln_eps, stream));
    if (struct_logits)
        TC_TRY(tc_head_logits_fwd(dt, w.mid, w.ld_mid, E, d->st_ln_g, d->st_ln_b, d->st_w, d->st_bias, struct_logits, M, E,
                                  d->n_struct, d->ln_eps, stream));
    return TC_OK;
}
