/*
 * transcodon_hip.h -- C ABI of libtranscodon_hip.so (MI355X / gfx950).
 *
 * The reference (guyuehuo/TransCodon) is pure Python/PyTorch and has no FFI of
 * its own; every entry point below therefore cites the reference Python
 * lines whose arithmetic it replaces.  The host-side mirror of the reference's
 * Python classes (transcodon_amd/) binds these symbols with ctypes.
 *
 * Conventions
 *  - plain pointers and sizes only; all pointers are DEVICE pointers unless a
 *    parameter says "host";
 *  - every function returns 0 on success or a negative TC_E* code and never
 *    throws; tc_last_error() gives the message of the calling thread's last
 *    failure;
 *  - nothing is allocated or freed on the device: scratch comes in through
 *    the caller's workspace (tc_workspace_bytes);
 *  - launches are asynchronous on `stream` (a hipStream_t passed as void*);
 *    calls on different streams may run concurrently as long as their output
 *    and workspace buffers do not overlap;
 *  - activations are row-major [rows, features]; token row index = b*T + t.
 */
#ifndef TRANSCODON_HIP_H
#define TRANSCODON_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Changes whenever a struct layout or a signature of this header changes or an entry point is added; a caller built against another
 * value must not call into the library.  tc_struct_bytes lets a binding that restates the structs (ctypes)
 * verify its own layout: which = 0 tc_layer_weights, 1 tc_encoder_desc, 2 tc_layer_grads, 3 tc_encoder_grads. */
#define TC_VERSION 131

/* status codes */
#define TC_OK 0
#define TC_EINVAL (-1)   /* bad shape / null pointer / unsupported size */
#define TC_ELAUNCH (-2)  /* HIP launch or runtime error */
#define TC_ENOSPC (-3)   /* workspace too small */

/* element types */
#define TC_F32 0
#define TC_BF16 1

/* GEMM epilogues */
#define TC_EPI_BIAS 0            /* out = A W^T + b                                  */
#define TC_EPI_BIAS_GELU 1       /* out = gelu_erf(A W^T + b)                        */
#define TC_EPI_BIAS_RESIDUAL 2   /* out_f32 = residual_f32 + A W^T + b               */
#define TC_EPI_QKV_ROPE 3        /* fused q/k/v projection, q scale, RoPE on q and k */
/* training-only epilogues (tc_gemm_train) */
#define TC_EPI_PLAIN_F32 4       /* out_f32 = A W^T (+ b if given); split-K partial slabs       */
#define TC_EPI_BIAS_GELU_SAVE 5  /* out2 = A W^T + b (kept for backward), out = gelu_erf(out2)  */
#define TC_EPI_DGELU 6           /* out = (A W^T) * gelu'(aux)   (gradient through the GELU)    */

int tc_version(void);
size_t tc_struct_bytes(int which);
const char *tc_last_error(void);

/*
 * Token + species embedding gather with pad zeroing.
 *   x[b*T+t, :] = (tok_table[tokens[b,t]] + species_table[species[b]]) * (tokens[b,t] != pad_idx)
 * Replaces transcodon/model.py:178-206 (embed_tokens, species_embedding added at
 * every position, multiplication by 1 - padding_mask).
 * tokens int32 [B,T]; species int32 [B]; tables f32; x f32 [B*T,E].
 * pad_flags (optional, may be NULL): uint8 [B*T], 1 where the token is padding.
 * Token / species ids out of range are reported through *bad_index (optional
 * device int32, set to 1) and produce a zero row instead of an out-of-bounds read.
 */
int tc_embed_fwd(const int32_t *tokens, const int32_t *species, const float *tok_table,
                 const float *species_table, float *x, uint8_t *pad_flags, int32_t *bad_index,
                 int B, int T, int E, int vocab, int n_species, int pad_idx, void *stream);

/*
 * Row LayerNorm with affine terms, biased variance, fp32 statistics.
 * Replaces torch.nn.LayerNorm(E, eps) as used by ESM1bLayerNorm
 * (transcodon/modules.py:27-41; call sites modules.py:87,100, model.py:232).
 * x f32 [M,E]; gamma/beta f32 [E]; out_f32 and/or out_bf16 [M,E] (either may be NULL).
 */
int tc_layernorm_fwd(const float *x, const float *gamma, const float *beta, float *out_f32,
                     void *out_bf16, int M, int E, float eps, void *stream);

/*
 * out[M,N] = epilogue(A[M,K] . W[N,K]^T + bias[N]) on the matrix cores.
 * dtype TC_BF16: A, W bf16, fp32 accumulate (mfma_f32_16x16x32_bf16);
 * dtype TC_F32 : A, W f32, exact f32 multiply-add chain (mfma_f32_16x16x4_f32).
 * Replaces the nn.Linear calls of the encoder: q/k/v/out projections
 * (transcodon/multihead_attention.py:262-265,399), fc1+gelu and fc2
 * (transcodon/modules.py:17-24,101-103), head dense layers (modules.py:155-156,173-174).
 *   TC_EPI_BIAS / TC_EPI_BIAS_GELU : out has element type `dtype`, leading dimension ldo.
 *   TC_EPI_BIAS_RESIDUAL           : out and residual are f32 [M,N] (may alias).
 *   TC_EPI_QKV_ROPE                : N = 3*E; columns [0,E) are q, [E,2E) k, [2E,3E) v;
 *       q = (A Wq^T + bq) * q_scale; q and k are rotated pairwise (2i,2i+1) inside each
 *       head of head_dim columns by the angle table rope_cos/rope_sin f32
 *       [T, head_dim/2] at position (row mod T)
 *       (multihead_attention.py:265,358-360; rotary-embedding-torch 0.8.6).
 * Requirements: N % 128 == 0, K % 64 == 0 (bf16) or K % 32 == 0 (f32), lda >= K, 16-byte
 * aligned pointers.
 * bf16 problems with N % 256 == 0 and at least one 256 x 256 tile per compute unit take the
 * large-tile kernels; there GELU is applied to the pre-activation rounded to bf16 (what a bf16
 * linear layer followed by GELU computes), and the q/k/v epilogue, when T % 256 == 0, also reads
 * rows 16 and 80 of the rotary tables.  A result whose rows are 6144 bytes apart stores slowly on
 * this memory system: prefer ldo = N + 64 for N = 3072 in bf16.
 */
int tc_gemm(int dtype, int epilogue, const void *A, int lda, const void *W, const float *bias,
            const float *residual, void *out, int ldo, int M, int N, int K, float q_scale,
            const float *rope_cos, const float *rope_sin, int T, int head_dim, void *stream);

/*
 * GEMM variants needed by the training step (pretraining.py:94-134 forward + autograd backward):
 *   TC_EPI_BIAS_GELU_SAVE  forward fc1: also keeps the pre-activation (out2, element type dtype);
 *   TC_EPI_DGELU           data gradient through fc1's GELU: aux = saved pre-activation [M,N];
 *                          out2 (optional, f32 [N]) += column sums of the result, i.e. fc1's bias
 *                          gradient (accumulated with float atomics: the last bits vary run to run);
 *   TC_EPI_PLAIN_F32       data / weight gradients with f32 output.  With splits > 1 the k range
 *                          is cut into `splits` parts and part s writes its partial product to
 *                          out + s*M*ldo (f32); tc_reduce_slabs adds the parts up.
 * Same shape rules as tc_gemm; weight gradients use it on transposed operands
 * (dW[N,K] = dY^T[N,M] . X^T[K,M]^T with the token dimension M as the reduction).
 */
int tc_gemm_train(int dtype, int epilogue, const void *A, int lda, const void *W, const float *bias,
                  const void *aux, void *out, void *out2, int ldo, int M, int N, int K, int splits,
                  void *stream);

/*
 * Key-validity bit masks for the attention kernel: bit j of words[b*nw + w] is 1
 * when key 64*w + j of sequence b exists (< T) and is not padding.
 * Replaces the key_padding_mask plumbing of transcodon/model.py:178,219-220 and
 * multihead_attention.py:373-379.  pad_flags uint8 [B,T]; words uint64 [B, ceil(T/64)].
 */
int tc_key_mask_words(const uint8_t *pad_flags, uint64_t *words, int B, int T, void *stream);

/*
 * Fused multi-head self-attention forward (no score matrix in memory):
 *   out[b,t,h*D:(h+1)*D] = softmax_fp32(q_h k_h^T masked with -inf at padded keys) v_h
 * with q already scaled and q,k already rotated.  Replaces
 * transcodon/multihead_attention.py:362-398 (bmm, masked_fill -inf, fp32 softmax, bmm,
 * head concat).  qkv [B*T, ldq] holds q at column 0, k at column E, v at column 2E
 * (E = H*D); out [B*T, E]; element type `dtype`.  key_words from tc_key_mask_words, or
 * NULL when no key is padded.  D must be 64; the bf16 kernels take T <= 8192 (forward and
 * backward) and reject longer sequences.
 */
int tc_attn_fwd(int dtype, const void *qkv, int ldq, const uint64_t *key_words, void *out, int B,
                int T, int H, int D, void *stream);
/*
 * The same product for q rows scaled by log2(e) / sqrt(D) instead of 1 / sqrt(D) (the q/k/v product's
 * q_scale argument; multihead_attention.py:350 scales by head_dim ** -0.5).  The scores then come out of
 * the matrix unit in base-2 units, the running softmax offset enters as the accumulator's start value and
 * exp2 applies to the accumulators directly: one multiply-add per score less.  bf16 only; the result is
 * the same softmax (the factor is a change of the exponential's base).  tc_encoder_forward uses it.
 */
int tc_attn_fwd_q2(int dtype, const void *qkv, int ldq, const uint64_t *key_words, void *out, int B, int T, int H,
                   int D, void *stream);

/*
 * Pre-softmax attention scores for need_head_weights=True callers:
 *   scores[b,h,i,j] = q_h[i] . k_h[j]   (-inf where key j is padding), f32 [B,H,T,T].
 * Replaces transcodon/multihead_attention.py:362-379,400-403.
 */
int tc_attn_scores(int dtype, const void *qkv, int ldq, const uint8_t *pad_flags, float *scores,
                   int B, int T, int H, int D, void *stream);

/*
 * Output heads after the dense+gelu GEMM: per row LayerNorm then a thin projection.
 *   out[m, v] = sum_e LN(g[m, col0 + e]) * W[v, e] + bias[v],  v < V <= 16
 * Replaces transcodon/modules.py:157-159 (RobertaLMHead, tied weight) and :175-177
 * (RNA_2D_RobertaLMHead; pass bias = out_proj.bias + bias).
 * g has element type `dtype`, leading dimension ldg; W f32 [V,E]; out f32 [M,V].
 */
int tc_head_logits_fwd(int dtype, const void *g, int ldg, int col0, const float *gamma,
                       const float *beta, const float *W, const float *bias, float *out, int M,
                       int E, int V, float eps, void *stream);

/*
 * Greedy synonymous-codon decode.  For amino acid i of sequence s the three logit rows
 * are logits[(row0[s] + 3*i + j) * ldl + 0..9], j = 0..2.  score(codon) = double product,
 * started at 1.0, of the fp32 softmax (over all 10 symbols) probabilities of its bases;
 * the first codon, in the reference's table order, with a strictly larger score wins.
 * Replaces the triple Python loop of transcodon/infer_utr.py:98-133.
 *   aa      uint8 [n_aa_total]   amino-acid codes: index into "FLIMVPTAY_HQNKDECWRSG"
 *   aa_off  int32 [n_seq+1]      prefix offsets of each sequence's residues in `aa`
 *   row0    int32 [n_seq]        logits row of the first CDS base of each sequence
 *   bases   uint8 [3*n_aa_total] output, ASCII 'A','U','C','G'
 */
int tc_codon_decode(const float *logits, int ldl, const uint8_t *aa, const int32_t *aa_off,
                    const int32_t *row0, uint8_t *bases, int n_seq, int n_aa_total, void *stream);

/*
 * Per-sequence integer statistics of decoded coding sequences (the output of tc_codon_decode, or any
 * ASCII A/C/G/T|U bytes), from which the reference's offline metrics follow exactly: GC content
 * (evaluation/gc.py:7-20), codon recovery against a second sequence set
 * (evaluation/codon_recover_rate.py:125-148) and the rare-codon share CFD (evaluation/cfd.py:9-44).
 *   bases      uint8 [3 * n_codons]   ref_bases  the same shape, or NULL
 *   codon_off  int32 [n_seq+1]        codons of sequence s: [codon_off[s], codon_off[s+1])
 *   codon_flags uint8 [64] or NULL    1 for every codon (index 16a+4b+c over A,C,G,T|U = 0..3) to count
 *   counts     int32 [n_seq,4]        {G+C letters, codons equal to ref, flagged codons, 1 if a letter
 *                                      outside ACGTU occurred}
 */
int tc_codon_stats(const uint8_t *bases, const uint8_t *ref_bases, const int32_t *codon_off,
                   const uint8_t *codon_flags, int32_t *counts, int n_seq, void *stream);

/*
 * %MinMax profile of decoded coding sequences (evaluation/minmax.py:9-79): for every codon position i of a
 * sequence with i + window <= its length, the mean actual / maximal / minimal / average usage frequency of the
 * window's codons and from them the signed percentage; the sums run in the host loop's order in f64, so the
 * values equal the reference's bit for bit.  Positions a window does not fit (the reference pads its list with
 * None there) hold NaN.  A window over amino acids that all have one codon divides by zero (the reference raises
 * ZeroDivisionError there) and holds the IEEE result.  A letter outside ACGTU or a codon absent from the table
 * ORs 1 into *status (zero it first) and leaves NaN in the windows that cover it.
 *   bases      uint8 [3 * n_codons]     codon_off int32 [n_seq+1] (as tc_codon_stats)
 *   tables     f64 [4,64]               rows: frequency of the codon, and the maximum, minimum and mean frequency
 *                                       among its amino acid's codons (index 16a+4b+c over A,C,G,T|U); NaN = absent
 *   out        f64 [n_codons]           status int32 [1]
 */
int tc_codon_minmax(const uint8_t *bases, const int32_t *codon_off, const double *tables, int window,
                    double *out, int32_t *status, int n_seq, long n_codons, void *stream);

/*
 * Mean cross-entropy with ignore_index = -100 over rows of thin logits, forward and
 * (optionally) gradient.  Replaces nn.CrossEntropyLoss(reduction='mean',
 * ignore_index=-100) of pretraining.py:54-55,109-112,124-127.
 *   logits f32 [M,V]; labels int32 [M]; loss_sum f32[1] and count int32[1] are ACCUMULATED
 *   (zero them first); dlogits f32 [M,V] (may be NULL) receives
 *   (softmax - onehot) * grad_scale for counted rows and 0 elsewhere.
 *   A label outside [0, V) other than -100 (torch raises on those) turns loss_sum into NaN.
 */
int tc_ce_loss_fwd_bwd(const float *logits, const int32_t *labels, float *loss_sum,
                       int32_t *count, float *dlogits, float grad_scale, int M, int V,
                       void *stream);

/*
 * Mean of the hidden states of each sequence over a row range (embedding extraction):
 *   pooled[b, :] = mean_{t in [start[b], start[b]+len[b])} rep[b*T + t, :]
 * Replaces the `repr_.mean(axis=1)` pooling of transcodon/infer_embedding.py:145 for
 * right-padded batches (only real, non-special positions are averaged).
 *   rep f32 [B*T,E]; start,len int32 [B]; pooled f32 [B,E]; len[b] == 0 gives a zero row.
 */
int tc_mean_pool(const float *rep, const int32_t *start, const int32_t *len, float *pooled, int B,
                 int T, int E, void *stream);

/* ---- backward-pass kernels of the training step (pretraining.py:94-134 + autograd) -- */

/*
 * out_t[n, m] = in[m, n] with element-type conversion (f32->f32, f32->bf16, bf16->bf16).
 * out_t has leading dimension ldt >= M; columns [M, ldt) are zero-filled, so the token
 * dimension can be the (padded) reduction dimension of a weight-gradient GEMM.
 * out_rowmajor (optional): converted row-major copy [M, ld_rm].  colsum (optional, f32 [N],
 * ACCUMULATED): column sums of `in`, i.e. the bias gradient of a Linear layer.
 */
int tc_transpose_cast(int in_dtype, int out_dtype, const void *in, int ld_in, void *out_t, int ldt,
                      void *out_rowmajor, int ld_rm, float *colsum, int M, int N, void *stream);

/*
 * Many dense transposes in one launch: dst[e] [cols, rows] = src[e] [rows, cols]^T for e < n (host arrays of device
 * pointers and sizes; rows and cols multiples of 64, 16-byte aligned, no overlap).  The backward pass transposes
 * every weight matrix this way once (the x.t() of F.linear's data gradient, taken by autograd for
 * modules.py:96-104, multihead_attention.py:139-148): one launch instead of one per matrix.
 */
int tc_transpose_batch(int dtype, int n, const void *const *src, void *const *dst, const int *rows,
                       const int *cols, void *stream);

/*
 * Weight-gradient product on row-major bf16 operands, no transposed copies:
 *   out[s][i, j] = sum over the s-th part of the rows m of A[m, i] * B[m, j]     (f32, s < splits)
 * A [M, lda] is the output gradient dY, B [M, ldb] the layer input X, so that summing the
 * `splits` slabs (tc_reduce_slabs) gives dW = dY^T X of a Linear layer.  I and J must be
 * multiples of 128; slab s starts at out + s*I*ldo.
 */
int tc_gemm_tn(const void *A, int lda, const void *B, int ldb, float *out, int ldo, int M, int I, int J,
               int splits, void *stream);

/*
 * colsum[n] += sum_m x[m, n] (f32, ACCUMULATED; the bias gradient of a Linear layer) for x of
 * type dtype; for f32 input an optional bf16 copy of x is written on the way (bf16_copy
 * [M, ld_copy]).  N must be a multiple of 8.
 */
int tc_colsum(int dtype, const void *x, int ld, void *bf16_copy, int ld_copy, float *colsum, int M, int N,
              void *stream);

/* out[i] = (accumulate ? out[i] : 0) + sum_s slabs[s*n + i]: combines split-K partial products. */
int tc_reduce_slabs(const float *slabs, float *out, int S, long n, int accumulate, void *stream);

/*
 * Backward of torch.nn.LayerNorm(E, eps): given the layer INPUT x (f32 or bf16, leading dimension
 * ldx), gamma and dy f32 [M,E]:  dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma,
 * written to or (accumulate != 0) added to dx f32 [M,E]; dgamma += sum_m dy * xhat,
 * dbeta += sum_m dy (both ACCUMULATED).
 */
int tc_layernorm_bwd(int x_dtype, const void *x, int ldx, const float *gamma, const float *dy, float *dx,
                     int accumulate, float *dgamma, float *dbeta, int M, int E, float eps, void *stream);
/* The same with the incoming gradient in either element type (bf16: what a bf16 Linear hands back
 * under autocast, transcodon/modules.py:83-105 run through pretraining.py:235) and its own row stride.
 * dx_bf16 (optional, dense [M,E]) receives the final dx rounded to bf16 and dx_colsum (optional, f32
 * [E], ACCUMULATED, needs dx_bf16) its column sums: the operand and the bias gradient of the Linear
 * layer below, which would otherwise take a tc_colsum pass over dx. */
int tc_layernorm_bwd_dy(int x_dtype, const void *x, int ldx, const float *gamma, int dy_dtype, const void *dy,
                        int lddy, float *dx, int accumulate, float *dgamma, float *dbeta, void *dx_bf16,
                        float *dx_colsum, int M, int E, float eps, void *stream);

/* out = dy * gelu'(u), elementwise over n values (exact erf form); u and out have type dtype. */
int tc_dgelu(int dtype, const float *dy, const void *u, void *out, long n, void *stream);

/*
 * Backward of tc_head_logits_fwd up to (not including) its LayerNorm: with ln = LN(g),
 *   d_ln[m,:] = sum_v dlogits[m,v] W[v,:]   (f32 [M,E], overwritten)
 *   dW[v,:]  += sum_m dlogits[m,v] ln[m,:]  (f32 [V,E], ACCUMULATED; the tied LM projection
 *                                            accumulates into the embed_tokens gradient)
 *   dbias[v] += sum_m dlogits[m,v]
 * Follow with tc_layernorm_bwd(x = g) to get d_g.  V must be 3 or 10, E <= 768.
 */
int tc_head_logits_bwd(int dtype, const void *g, int ldg, int col0, const float *gamma, const float *beta,
                       const float *W, const float *dlogits, float *d_ln, float *dW, float *dbias, int M,
                       int E, int V, float eps, void *stream);

/*
 * Backward of tc_embed_fwd: d_tok_table[tokens[b,t]] += dx[b*T+t] and
 * d_species_table[species[b]] += sum_t dx[b*T+t], over non-pad positions (both ACCUMULATED).
 */
int tc_embed_bwd(const int32_t *tokens, const int32_t *species, const float *dx, float *d_tok_table,
                 float *d_species_table, int B, int T, int E, int vocab, int n_species, int pad_idx,
                 void *stream);

/*
 * Backward of the rotary rotation and the q scale, in place on the q ([0,E)) and k ([E,2E))
 * column sections of d_qkv [M, ld]: (d0, d1) <- (d0 c + d1 s, d1 c - d0 s) * (q ? q_scale : 1).
 */
int tc_rope_bwd(int dtype, void *dqkv, int ld, const float *rope_cos, const float *rope_sin, int M, int T,
                int E, int head_dim, float q_scale, void *stream);

/*
 * tc_attn_fwd that also returns lse[b,h,t] = log sum_k exp(score[t,k]) (natural log, f32 [B,H,T];
 * +inf for a query without any valid key), the only softmax state the backward needs.
 */
int tc_attn_fwd_train(int dtype, const void *qkv, int ldq, const uint64_t *key_words, void *out, float *lse,
                      int B, int T, int H, int D, void *stream);

/*
 * Backward of tc_attn_fwd, i.e. of transcodon/multihead_attention.py:362-398 as autograd would
 * differentiate it: given qkv (as passed to the forward), out, lse and d_out [B*T, E],
 *   d_qkv[:, 0:E] = dS K,  d_qkv[:, E:2E] = dS^T Q,  d_qkv[:, 2E:3E] = P^T d_out
 * with P = exp(S - lse) (0 at padded keys) and dS = P o (d_out V^T - delta),
 * delta[b,h,t] = sum_d d_out o out (f32 [B,H,T] scratch supplied by the caller, overwritten).
 * Without rope tables the gradients are with respect to the scaled, rotated q and rotated k.
 * With rope_cos / rope_sin (f32 [T, D/2], as for TC_EPI_QKV_ROPE) the rotation and the q scale are
 * undone on the way out (what tc_rope_bwd does), so d_qkv is the gradient of the q/k/v projection
 * outputs.  d_qkv [B*T, ldg] has the element type `dtype`.  D must be 64.
 */
int tc_attn_bwd(int dtype, const void *qkv, int ldq, const uint64_t *key_words, const void *out,
                const void *d_out, const float *lse, float *delta, void *d_qkv, int ldg,
                const float *rope_cos, const float *rope_sin, float q_scale, int B, int T, int H, int D,
                void *stream);

/*
 * tc_attn_bwd without a scratch argument, for callers that cannot allocate device memory (code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the
 * reference.  delta lives in a 16 MiB array built into the library (one copy per GPU, held until the
 * process ends): B*H*T must be <= 4 194 304 or the call fails with TC_EINVAL.  The call allocates and
 * frees nothing.  All calls on one GPU share the array: calls that overlap on one GPU (different
 * streams or threads) race; calls in stream order on one stream do not.  Results equal tc_attn_bwd's.
 */
/* Resolves, for the CURRENT device, the address of that array (hipGetSymbolAddress).  The first
 * tc_attn_bwd_static call on a device does the same by itself, but that first call must then happen
 * outside stream capture; a caller that captures graphs calls tc_init_device once per device beforehand.
 * No counterpart in the reference. */
int tc_init_device(void);
int tc_attn_bwd_static(int dtype, const void *qkv, int ldq, const uint64_t *key_words, const void *out,
                       const void *d_out, const float *lse, void *d_qkv, int ldg, const float *rope_cos,
                       const float *rope_sin, float q_scale, int B, int T, int H, int D, void *stream);

/* ---- LayerNorm folded into the products on either side of it (bf16, large batches) ----- */
/*
 * transcodon/modules.py:87,100: h = LayerNorm(x) feeds a Linear layer (q/k/v at :88-95 through
 * multihead_attention.py:260-262, fc1 at :101), and x itself is the result of the Linear layer +
 * residual add before it (:96-97 out_proj, :102-104 fc2).  With xb = bf16(x), mean / rstd of xb's row,
 *     Linear(LayerNorm(xb))[n] = rstd * sum_k xb[k] (gamma o W)[n,k] - rstd * mean * s[n] + c[n],
 *     s[n] = sum_k (gamma o W)[n,k],   c[n] = sum_k beta[k] W[n,k] + b[n],
 * so the normalised activations never exist in memory: the residual product writes xb and the row sums
 * next to x, the next product multiplies xb by gamma o W and its epilogue applies the rest.  The
 * LayerNorm input is rounded to bf16 (the separate kernel rounds its output instead); everything
 * else is f32.  Only the large-tile bf16 kernels have these epilogues: ask tc_ln_fold_supported.
 *
 * tc_ln_fold_weights   W f32 [N,K], gamma/beta f32 [K], bias f32 [N] or NULL -> Wf bf16 [N,K] =
 *                      bf16(gamma o W), s f32 [N] (sums of the ROUNDED Wf rows), c f32 [N].
 * tc_embed_fwd_ln      tc_embed_fwd that also writes xb bf16 [B*T,E] and ln_rows f32 [B*T,2] =
 *                      (rstd, -rstd * mean) of xb's rows; optionally x_lo bf16 [B*T,E] = bf16(x - xb), the
 *                      remainder plane of the split residual stream (x may then be NULL).
 * Split residual stream: tc_encoder_forward keeps x as the plane pair (xb, x_lo), x = xb + x_lo to 16 bits
 *                      of mantissa (error 7e-6 of a row's standard deviation after 24 residual adds, 200
 *                      times below the bf16 rounding of xb that the products see anyway), so that the residual
 *                      products move the same bytes as with an f32 stream and the hi plane IS the operand of
 *                      the next product: no copy is written.
 * tc_gemm_residual_split  x += A W^T + b on the plane pair in place, plus the row sums of the new hi plane
 *                      (row_acc as for tc_gemm_residual_ln).  From K = 768 on (twelve 64-deep k-tiles or more) the
 *                      planes are not loaded ahead of the k-loop but streamed in under it and added to the
 *                      accumulators by the matrix unit (1.0 x bf16 products, exact), the bias last: the f32 sum is
 *                      then rounded in another order than tc_gemm(TC_EPI_BIAS_RESIDUAL) rounds it (the pair agrees
 *                      with that product to 2^-15 of the value plus 2^-16 absolute, the hi plane differs from its
 *                      bf16 rounding in fewer than 2 elements in 1000, by one bf16 step).  Environment
 *                      TC_RES_STREAM=0 keeps the load-ahead form for every K, =2 streams two plane pieces per k-tile
 *                      instead of three (another rounding order within the same bound).  M * ldo * 2 must stay below 4 GiB.
 * tc_layernorm_fwd_split  tc_layernorm_fwd reading the plane pair;  tc_split_to_f32: out[i] = hi[i] + lo[i].
 * tc_gemm_residual_ln  tc_gemm(TC_EPI_BIAS_RESIDUAL) that also writes out_bf16 [M,ldo] = bf16(out) and
 *                      adds, per row, the sum and the sum of squares of the bf16-rounded results to
 *                      row_acc (uint64 [rows,2], two's-complement integers in units of 2^-24: integer
 *                      adds, so the result does not depend on the order of the column tiles).  row_acc
 *                      must be zero before the call.
 * tc_ln_rows_finalize  row_acc -> ln_rows f32 [M,2] = (rstd, -rstd * mean) over E columns, in double
 *                      precision; clears row_acc for the next product.
 * tc_gemm_ln           tc_gemm(TC_EPI_QKV_ROPE or TC_EPI_BIAS_GELU) on A = xb with Wf, c, s and ln_rows.
 *                      ln_rows must be readable up to the next multiple of 256 rows.
 * tc_ln_fold_supported 1 if a forward of B x T tokens with widths E, F runs all its products on the
 *                      kernels that have these epilogues (then tc_encoder_forward folds when the
 *                      descriptor's layers carry the folded weights), else 0.
 */
int tc_ln_fold_weights(const float *W, const float *gamma, const float *beta, const float *bias, void *Wf,
                       float *s_out, float *c_out, int N, int K, void *stream);
int tc_embed_fwd_ln(const int32_t *tokens, const int32_t *species, const float *tok_table,
                    const float *species_table, float *x, uint8_t *pad_flags, int32_t *bad_index, int B, int T,
                    int E, int vocab, int n_species, int pad_idx, void *xb, void *x_lo, float *ln_rows, float eps,
                    void *stream);
int tc_gemm_residual_split(int dtype, const void *A, int lda, const void *W, const float *bias, void *x_hi, void *x_lo,
                           unsigned long long *row_acc, int ldo, int M, int N, int K, void *stream);
int tc_layernorm_fwd_split(const void *x_hi, const void *x_lo, const float *gamma, const float *beta, float *out_f32,
                           void *out_bf16, int M, int E, float eps, void *stream);
int tc_split_to_f32(const void *x_hi, const void *x_lo, float *out, long n, void *stream);
int tc_gemm_residual_ln(int dtype, const void *A, int lda, const void *W, const float *bias, const float *residual,
                        void *out, void *out_bf16, unsigned long long *row_acc, int ldo, int M, int N, int K,
                        void *stream);
int tc_ln_rows_finalize(unsigned long long *row_acc, float *ln_rows, int M, int E, float eps, void *stream);
/* dst[0..n) = src[0..n) (f32, n a multiple of 4, 16-byte aligned) as a kernel launch: what the forward copies
 * representations with.  On this stack the memset and copy nodes of a captured graph are not ordered against the
 * graph's previous launch, kernel nodes are. */
int tc_copy_f32(float *dst, const float *src, long n, void *stream);
/* x[m,n] += bias[n] + sum_s slabs[s][m][n] (f32; slabs = S split-K partial products of tc_gemm_train's plain
 * epilogue, M*N elements apart): the bias + residual epilogue (modules.py:96-97,102-104) of a product too small
 * to fill the chip with whole-K tiles.  tc_encoder_forward uses it for the f32 out-proj and fc2 of short batches. */
int tc_splitk_bias_residual(const float *slabs, int S, const float *bias, float *x, int M, int N, void *stream);
/* row_acc [M,2] to zero, as a kernel launch (what a forward starts with; graph-capture safe) */
int tc_ln_rows_clear(unsigned long long *row_acc, int M, void *stream);
int tc_gemm_ln(int dtype, int epilogue, const void *A, int lda, const void *Wf, const float *c, const float *s,
               const float *ln_rows, void *out, int ldo, int M, int N, int K, float q_scale, const float *rope_cos,
               const float *rope_sin, int T, int head_dim, void *stream);
int tc_ln_fold_supported(int dtype, int B, int T, int E, int F);
/* the same for the products without a rotary epilogue only (out-proj, fc1, fc2) on M rows: what the packed
 * forward asks before it folds the second LayerNorm of every layer and keeps the residual stream as planes */
int tc_ln_fold_rows_supported(int dtype, long M, int E, int F);

/* ---- whole-encoder convenience (one call per forward) ---------------------------- */

typedef struct tc_layer_weights {
    const float *ln1_g, *ln1_b;   /* layers.i.self_attn_layer_norm                    */
    const void *w_qkv;            /* [3E,E] rows: q_proj, k_proj, v_proj (dtype)      */
    const float *b_qkv;           /* [3E]                                             */
    const void *w_o;              /* [E,E]  self_attn.out_proj                        */
    const float *b_o;
    const float *ln2_g, *ln2_b;   /* layers.i.final_layer_norm                        */
    const void *w_fc1;            /* [F,E]                                            */
    const float *b_fc1;
    const void *w_fc2;            /* [E,F]                                            */
    const float *b_fc2;
    const float *rope_cos, *rope_sin; /* f32 [T_table, D/2] for this layer's rope.freqs */
    /* Optional (all six or none, bf16 only): the two LayerNorms folded into the products that follow
     * them, as tc_ln_fold_weights makes them from (w_qkv, ln1_g, ln1_b, b_qkv) and (w_fc1, ln2_g,
     * ln2_b, b_fc1).  When every layer has them and tc_ln_fold_supported says so, tc_encoder_forward
     * runs no LayerNorm kernel inside the layers. */
    const void *w_qkv_ln;         /* [3E,E] bf16(gamma1 o W)                          */
    const float *c_qkv_ln, *s_qkv_ln; /* [3E] beta1 W^T + b;  row sums of w_qkv_ln     */
    const void *w_fc1_ln;         /* [F,E]                                            */
    const float *c_fc1_ln, *s_fc1_ln;
} tc_layer_weights;

typedef struct tc_encoder_desc {
    int32_t dtype;                /* TC_BF16 or TC_F32: element type of GEMM operands  */
    int32_t num_layers, E, F, H, vocab, n_species, n_struct, pad_idx;
    int32_t rope_table_len;       /* rows in every rope_cos/rope_sin table (>= T)      */
    float ln_eps;
    const float *tok_table;       /* embed_tokens.weight [vocab,E] f32                 */
    const float *species_table;   /* species_embedding.weight [n_species,E] f32        */
    const tc_layer_weights *layers; /* host array [num_layers]                          */
    const float *lnf_g, *lnf_b;   /* emb_layer_norm_after                              */
    const void *w_heads;          /* [2E,E] rows: lm_head.dense, structure_linear.dense */
    const float *b_heads;         /* [2E]                                              */
    const float *lm_ln_g, *lm_ln_b;      /* lm_head.layer_norm                          */
    const float *lm_bias;                /* lm_head.bias [vocab] (weight = tok_table)   */
    const float *st_ln_g, *st_ln_b;      /* structure_linear.layer_norm                 */
    const float *st_w;                   /* structure_linear.out_proj.weight [n_struct,E] */
    const float *st_bias;                /* out_proj.bias + structure_linear.bias        */
    /* Optional (all three or none, bf16 only): emb_layer_norm_after folded into the two head dense layers, as
     * tc_ln_fold_weights makes them from (w_heads in f32, lnf_g, lnf_b, b_heads).  Used by a folded forward
     * (see tc_layer_weights) that is not asked for rep_last: the last fc2's row sums then serve the head product
     * and no LayerNorm kernel runs at all (model.py:232-240 become part of the [2E,E] product). */
    const void *w_heads_ln;              /* [2E,E] bf16(gamma o W)                       */
    const float *c_heads_ln, *s_heads_ln; /* [2E]                                        */
} tc_encoder_desc;

/* Bytes of device workspace tc_encoder_forward needs for B sequences of T tokens. */
size_t tc_workspace_bytes(const tc_encoder_desc *desc, int B, int T);

/*
 * ProteinBertModel.forward (transcodon/model.py:175-259) for a padded batch:
 * embedding -> num_layers x pre-LN TransformerLayer (modules.py:83-105) -> final
 * LayerNorm -> LM head and structure head.
 *   tokens int32 [B,T], species int32 [B]                       (device)
 *   logits f32 [B*T,vocab]; struct_logits f32 [B*T,n_struct] or NULL;
 *   rep_last f32 [B*T,E] or NULL  (post emb_layer_norm_after states, model.py:232-237);
 *   rep_layer / rep_out: optional host arrays of n_rep layer indices (0..num_layers-1,
 *   0 = embedding output) and matching device f32 [B*T,E] buffers for
 *   representations of other layers (model.py:210-211,226-227).
 *   status int32 [1] device (optional): set to 1 if a token/species id was out of range.
 * desc and desc->layers are read on the host during the call only.
 */
int tc_encoder_forward(const tc_encoder_desc *desc, const int32_t *tokens, const int32_t *species,
                       int B, int T, float *logits, float *struct_logits, float *rep_last,
                       int n_rep, const int32_t *rep_layer, float *const *rep_out,
                       int32_t *status, void *workspace, size_t workspace_bytes, void *stream);

/*
 * Packed (variable-length) batches: n_seq sequences lie one after the other in the row dimension,
 * sequence i in rows [cu[i], cu[i+1]); no padding rows exist, so GEMM / LayerNorm / head rows are
 * sum(T_i) instead of B * T_max.  This is how the embedding extraction over coding sequences of very
 * different lengths (transcodon/infer_embedding.py:17-148, one sequence per forward there) is batched.
 *
 * tc_gemm_qkv_packed   TC_EPI_QKV_ROPE where row m is rotated by table row pos[m] (int32 [M], device;
 *                      pos[m] = m - cu[i]: positions restart per sequence, multihead_attention.py:358-360
 *                      applied to each sequence on its own).  Every pos[m] must be < table_rows.
 * tc_attn_fwd_packed   tc_attn_fwd per (sequence, query block): blocks is a device array of n_blocks
 *                      entries {first row of the sequence, its length, query block index, 0} (int32 x 4),
 *                      query blocks of tc_attn_packed_block_rows(dtype) rows, in any order; max_len = the
 *                      longest sequence; flops feeds the timers only (4 H D sum T_i^2, or 0).
 * tc_mean_pool_packed  tc_mean_pool with absolute first rows.
 * tc_encoder_forward_packed   tc_encoder_forward on such a batch.  tokens / species_row / pos int32 [M]
 *                      (species_row[m] = species id of row m's sequence); cu_seqlens_host is a HOST array
 *                      [n_seq+1] used to validate lengths against the rope table and the block list.
 *                      logits [M,vocab] or NULL (then the heads are skipped: embedding extraction);
 *                      struct_logits [M,n_struct] or NULL; rep_last [M,E] or NULL.
 *                      row_panels (or NULL): device int32 [n_panels,4], every sequence cut into pieces of at
 *                      most 256 rows {first row, end row, position of the first row, 0}, n_panels =
 *                      sum ceil(T_i / 256); with it a large bf16 batch runs the q/k/v product in its folded
 *                      form as well (tc_gemm_ln_qkv_panels) and no LayerNorm kernel inside the layers.
 *                      Workspace: tc_workspace_bytes(desc, 1, M).
 * tc_gemm_ln_qkv_panels  tc_gemm_ln with the q/k/v epilogue for a packed batch: row panel i of the launch is
 *                      rows [panels[4i], panels[4i+1]) of A and out, all of one sequence, rotary positions
 *                      panels[4i+2] + r (tables of table_rows >= 96 rows).  Rows and positions in the list
 *                      are the caller's contract (not checked on the device).
 */
int tc_gemm_qkv_packed(int dtype, const void *A, int lda, const void *W, const float *bias, void *out, int ldo,
                       int M, int N, int K, float q_scale, const float *rope_cos, const float *rope_sin,
                       const int32_t *pos, int table_rows, int head_dim, void *stream);
int tc_gemm_ln_qkv_panels(int dtype, const void *A, int lda, const void *Wf, const float *c, const float *s,
                          const float *rows, void *out, int ldo, int M, int N, int K, float q_scale,
                          const float *rope_cos, const float *rope_sin, int table_rows, int head_dim,
                          const int32_t *panels, int n_panels, void *stream);
int tc_attn_packed_block_rows(int dtype);
int tc_attn_fwd_packed(int dtype, const void *qkv, int ldq, const int32_t *blocks, int n_blocks, int max_len,
                       void *out, int H, int D, double flops, void *stream);
/* tc_attn_fwd_packed for q rows scaled by log2(e) / sqrt(D) (see tc_attn_fwd_q2); bf16 only */
int tc_attn_fwd_packed_q2(int dtype, const void *qkv, int ldq, const int32_t *blocks, int n_blocks, int max_len,
                          void *out, int H, int D, double flops, void *stream);
int tc_mean_pool_packed(const float *rep, const int32_t *row_start, const int32_t *len, float *pooled, int n_seq,
                        int total_rows, int E, void *stream);
int tc_encoder_forward_packed(const tc_encoder_desc *desc, const int32_t *tokens, const int32_t *species_row,
                              const int32_t *pos, const int32_t *attn_blocks, int n_blocks,
                              const int32_t *row_panels, int n_panels,
                              const int32_t *cu_seqlens_host, int n_seq, float *logits, float *struct_logits,
                              float *rep_last, int32_t *status, void *workspace, size_t workspace_bytes,
                              void *stream);

/* ---- per-kernel timing for roofline reports -------------------------------------- */
#define TC_PROF_GEMM 0
#define TC_PROF_ATTN 1
#define TC_PROF_LAYERNORM 2
#define TC_PROF_EMBED 3
#define TC_PROF_HEAD 4
#define TC_PROF_DECODE 5
#define TC_PROF_LN_ROWS 6      /* tc_ln_rows_finalize */
#define TC_PROF_NKINDS 7
/*
 * When enabled, every launch of the selected kernel families carries a start and a stop HIP
 * event on its own dispatch (hipExtLaunchKernelGGL: the kernel's own timestamps, no marker
 * packets on the stream).  tc_prof_collect synchronises those events and returns, per
 * kind, the launch count, the summed device time in milliseconds and the summed
 * algorithmic work (flops for GEMM/ATTN, bytes for the others) since the last reset.
 */
int tc_prof_enable(int enabled);
int tc_prof_reset(void);
int tc_prof_collect(int kind, int64_t *launches, double *ms_total, double *work_total);

/* ---- training step (pretraining.py:94-134: forward, loss, autograd backward, AdamW) ------ */

/*
 * Gradient buffers, f32, shaped like the weights of tc_layer_weights / tc_encoder_desc (the
 * packed [3E,E] q/k/v matrix and [2E,E] head matrix included).  Every buffer is ACCUMULATED
 * into: zero them before the first micro-batch of an optimiser step.
 */
typedef struct tc_layer_grads {
    float *ln1_g, *ln1_b, *w_qkv, *b_qkv, *w_o, *b_o, *ln2_g, *ln2_b, *w_fc1, *b_fc1, *w_fc2, *b_fc2;
} tc_layer_grads;

typedef struct tc_encoder_grads {
    float *tok_table;       /* embed_tokens.weight: embedding + tied LM projection */
    float *species_table;
    tc_layer_grads *layers;
    float *lnf_g, *lnf_b;
    float *w_heads, *b_heads;
    float *lm_ln_g, *lm_ln_b, *lm_bias;
    float *st_ln_g, *st_ln_b, *st_w;
    float *st_bias;         /* gradient of both structure_linear.out_proj.bias and .bias */
} tc_encoder_grads;

/* Bytes of workspace for tc_encoder_forward_train + tc_encoder_backward on a [B,T] batch. */
size_t tc_train_workspace_bytes(const tc_encoder_desc *desc, int B, int T);

/*
 * tc_encoder_forward (transcodon/model.py:175-259) that keeps the activations the backward
 * needs in `workspace`, which the caller must leave untouched until tc_encoder_backward has
 * been enqueued.  Both heads are always evaluated.
 */
int tc_encoder_forward_train(const tc_encoder_desc *desc, const int32_t *tokens, const int32_t *species,
                             int B, int T, float *logits, float *struct_logits, int32_t *bad_index,
                             void *workspace, size_t workspace_bytes, void *stream);

/*
 * Gradients of every parameter given d(loss)/d(logits) [B*T,vocab] and, optionally,
 * d(loss)/d(struct_logits) [B*T,n_struct] (NULL when the structure term is absent): what
 * loss.backward() produces in CodonModel.training_step (pretraining.py:94-134).
 */
int tc_encoder_backward(const tc_encoder_desc *desc, const tc_encoder_grads *grads, const int32_t *tokens,
                        const int32_t *species, int B, int T, const float *dlogits,
                        const float *dstruct_logits, void *workspace, size_t workspace_bytes, void *stream);

/*
 * tc_encoder_backward with a host callback for overlapping the data-parallel gradient exchange
 * with the rest of the backward (what DDP's per-bucket hooks do for pretraining.py:243):
 * ready(l, user) is called on the calling thread as soon as every kernel that writes the
 * gradients of layer l (l = num_layers-1 ... 0) has been enqueued on `stream`, and ready(-1, user)
 * once after the embedding / final LayerNorm / head gradients.  The callee typically records an
 * event on `stream` and starts an all-reduce of that layer's gradient range on another stream.
 */
typedef void (*tc_grads_ready_fn)(int layer, void *user);
int tc_encoder_backward_notify(const tc_encoder_desc *desc, const tc_encoder_grads *grads, const int32_t *tokens,
                               const int32_t *species, int B, int T, const float *dlogits,
                               const float *dstruct_logits, void *workspace, size_t workspace_bytes,
                               tc_grads_ready_fn ready, void *user, void *stream);

/*
 * The objective of pretraining.py:109-130 with its gradient:
 *   loss = CE_mean(logits, labels; ignore -100)
 *          + struct_weight * CE_mean(struct_logits, struct_labels; ignore -100)   if any struct label is set
 * loss f32[1] (device).  dlogits / dstruct_logits (optional) receive d(upstream * loss)/d(logits).
 * stats: 16 bytes of device scratch (sums and counts).  struct_* may all be NULL.
 */
int tc_mlm_loss(const float *logits, const int32_t *labels, const float *struct_logits,
                const int32_t *struct_labels, float *loss, float *dlogits, float *dstruct_logits, float *stats,
                float struct_weight, float upstream, int M, int V, int S, void *stream);

/*
 * Masked-LM corruption of a right-padded batch on the device, the arithmetic of DataCollator._mask_seq
 * and DataPreprocessor (transcodon/pipeline.py:167-195, 343-361): per sequence of n = lens[b] tokens,
 * num_changed = int(n * mask_proportion) distinct positions are drawn uniformly from 1..n-2; of these
 * int(num_changed * mask_percent) become mask_idx, int(num_changed * leave_percent) keep their token and
 * the rest get a uniformly drawn id from coding_ids.  labels = original token at the drawn positions,
 * -100 elsewhere; struct_labels (optional, in place) is set to -100 outside the drawn positions.
 * The reference draws from numpy's global generator; this uses a counter-based generator keyed by
 * (seed, sequence, position), so the two agree in distribution, not draw by draw.
 */
int tc_mlm_mask(const int32_t *tokens, const int32_t *lens, int32_t *tokens_out, int32_t *labels,
                int32_t *struct_labels, int B, int T, uint64_t seed, double mask_proportion, double mask_percent,
                double leave_percent, int mask_idx, const int32_t *coding_ids, int n_coding, void *stream);

/*
 * Random crop and right-padding of an int batch on the device: what DataTrimmer and DataPadder
 * (transcodon/pipeline.py:250-263, 300-309) do with token strings.  Row b keeps
 * keep = min(lens[b], T_out) entries starting at start[b] (clamped to [0, lens[b] - keep]; NULL = 0);
 * out[b, t] = in[b, start[b] + t] for t < keep, `fill` after it (pad_idx for tokens, -100 for labels);
 * lens_out[b] (optional) = keep.  in [B,T_in], out [B,T_out], distinct buffers.
 */
int tc_crop_pad(const int32_t *in, int32_t *out, const int32_t *start, const int32_t *lens, int32_t *lens_out,
                int B, int T_in, int T_out, int fill, void *stream);

/*
 * One AdamW update over n contiguous f32 values, torch.optim.AdamW semantics (pretraining.py:64-70:
 * decoupled weight decay on every parameter): p *= 1 - lr*wd; m, v moments of grad*grad_scale;
 * p -= lr/(1-b1^step) * m / (sqrt(v)/sqrt(1-b2^step) + eps).  step counts from 1.
 */
int tc_adamw_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, long n, float lr,
                  float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                  void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TRANSCODON_HIP_H */
