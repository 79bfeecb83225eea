// HBM-bound row kernels: embedding gather, LayerNorm, output heads, codon decode, CE loss.
// One 64-lane wave owns one row; lanes read 16 bytes at a time; reductions are
// wave-wide shuffles, no LDS.
#include "common.h"
#include <type_traits>
#include <stdlib.h>

namespace {

// ---- embedding ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_kernel(const int32_t *__restrict__ tokens,
                                                    const int32_t *__restrict__ species,
                                                    const float *__restrict__ tok_table,
                                                    const float *__restrict__ sp_table,
                                                    float *__restrict__ x, uint8_t *pad_flags,
                                                    int32_t *bad_index, int M, int T, int E,
                                                    int vocab, int n_species, int pad_idx,
                                                    uint16_t *__restrict__ xb, float *__restrict__ ln_rows,
                                                    float eps, uint16_t *__restrict__ x_lo) {
    // xb / ln_rows (optional, together): bf16 copy of the row and (rstd, -rstd * mean) of that copy,
    // for a first LayerNorm folded into the product that follows.  x_lo (optional, with xb): the
    // remainder plane bf16(x - xb) of the split residual stream; x itself may then be NULL.
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int tok = tokens[row];
    const int sp = species[row / T];
    const bool bad = tok < 0 || tok >= vocab || sp < 0 || sp >= n_species;
    const bool pad = tok == pad_idx;
    if (lane == 0) {
        if (pad_flags) pad_flags[row] = pad ? 1 : 0;
        if (bad && bad_index) *bad_index = 1;
    }
    float *dst = x + (long)row * E;
    if (pad || bad) {
        for (int c = lane * 4; c < E; c += 256) {
            if (x) *reinterpret_cast<f32x4 *>(dst + c) = f32x4{0, 0, 0, 0};
            if (xb) *reinterpret_cast<u32x2 *>(xb + (long)row * E + c) = u32x2{0, 0};
            if (x_lo) *reinterpret_cast<u32x2 *>(x_lo + (long)row * E + c) = u32x2{0, 0};
        }
        if (ln_rows && lane == 0) *reinterpret_cast<f32x2 *>(ln_rows + 2L * row) = f32x2{rsqrtf(eps), 0.f};
        return;
    }
    const float *a = tok_table + (long)tok * E;
    const float *b = sp_table + (long)sp * E;
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane * 4; c < E; c += 256) {
        const f32x4 va = *reinterpret_cast<const f32x4 *>(a + c);
        const f32x4 vb = *reinterpret_cast<const f32x4 *>(b + c);
        const f32x4 v = va + vb;
        if (x) *reinterpret_cast<f32x4 *>(dst + c) = v;
        if (xb) {
            u32x2 w;
            w[0] = tc_pack_bf16(v[0], v[1]);
            w[1] = tc_pack_bf16(v[2], v[3]);
            *reinterpret_cast<u32x2 *>(xb + (long)row * E + c) = w;
            const float p = tc_bf16_lo(w[0]), q = tc_bf16_hi(w[0]), r = tc_bf16_lo(w[1]), t = tc_bf16_hi(w[1]);
            if (x_lo) {
                u32x2 l;
                l[0] = tc_pack_bf16(v[0] - p, v[1] - q);
                l[1] = tc_pack_bf16(v[2] - r, v[3] - t);
                *reinterpret_cast<u32x2 *>(x_lo + (long)row * E + c) = l;
            }
            s1 += (p + q) + (r + t);
            s2 += (p * p + q * q) + (r * r + t * t);
        }
    }
    if (ln_rows) {
        s1 = tc_wave_sum(s1);
        s2 = tc_wave_sum(s2);
        if (lane == 0) {
            const double mean = (double)s1 / E;
            double var = (double)s2 / E - mean * mean;
            var = var > 0.0 ? var : 0.0;
            const float rstd = (float)(1.0 / sqrt(var + (double)eps));
            *reinterpret_cast<f32x2 *>(ln_rows + 2L * row) = f32x2{rstd, (float)(-(double)rstd * mean)};
        }
    }
}

// ---- LayerNorm folded into the neighbouring products: row statistics and weight preparation ------
// The residual product leaves, per row, the exact integer sums (units of 2^-24) of its bf16-rounded
// results and of their squares; this turns them into (rstd, -rstd * mean) and clears the
// accumulators for the next product.
__global__ __launch_bounds__(256) void ln_rows_finalize_kernel(unsigned long long *__restrict__ row_acc,
                                                               float *__restrict__ ln_rows, int M, int E, float eps) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= M) return;
    const double scale = 1.0 / 16777216.0;
    const double s1 = (double)(long long)row_acc[2L * row] * scale, s2 = (double)(long long)row_acc[2L * row + 1] * scale;
    row_acc[2L * row] = 0;
    row_acc[2L * row + 1] = 0;
    const double mean = s1 / E;
    double var = s2 / E - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    *reinterpret_cast<f32x2 *>(ln_rows + 2L * row) = f32x2{rstd, (float)(-(double)rstd * mean)};
}

// Row accumulators to zero.  A kernel, not hipMemsetAsync: in a captured graph the memset node is not ordered
// against the previous launch of the same graph, so back-to-back replays cleared the sums of the forward
// still running.
__global__ __launch_bounds__(256) void ln_rows_clear_kernel(unsigned long long *__restrict__ row_acc, long n2) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n2; i += (long)gridDim.x * 256)
        *reinterpret_cast<u32x4 *>(row_acc + 2 * i) = u32x4{0u, 0u, 0u, 0u};
}

// x[m,n] += bias[n] + sum over S split-K partial products [S][M][N]: the residual epilogue of a small f32 product
// whose k range was cut into S parts to fill the chip (the parts are added in index order: deterministic)
__global__ __launch_bounds__(256) void splitk_bias_residual_kernel(const float *__restrict__ slabs, int S, long slab_elems,
                                                                   const float *__restrict__ bias, float *__restrict__ x,
                                                                   long n4, int N) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        f32x4 acc = *reinterpret_cast<const f32x4 *>(slabs + 4 * i);
        for (int sl = 1; sl < S; ++sl) acc += *reinterpret_cast<const f32x4 *>(slabs + sl * slab_elems + 4 * i);
        const int n = (int)((4 * i) % N);
        acc += *reinterpret_cast<const f32x4 *>(bias + n);
        *reinterpret_cast<f32x4 *>(x + 4 * i) = *reinterpret_cast<const f32x4 *>(x + 4 * i) + acc;
    }
}

// Device-to-device copy of f32 data as a kernel (for the same reason: copy nodes of a captured graph)
__global__ __launch_bounds__(256) void copy_f32_kernel(const float *__restrict__ src, float *__restrict__ dst, long n4) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256)
        *reinterpret_cast<f32x4 *>(dst + 4 * i) = *reinterpret_cast<const f32x4 *>(src + 4 * i);
}

// x = hi + lo of the split residual stream as f32 (representations of inner layers)
__global__ __launch_bounds__(256) void split_to_f32_kernel(const uint16_t *__restrict__ hi, const uint16_t *__restrict__ lo,
                                                           float *__restrict__ out, long n4) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const u32x2 h = *reinterpret_cast<const u32x2 *>(hi + 4 * i), l = *reinterpret_cast<const u32x2 *>(lo + 4 * i);
        *reinterpret_cast<f32x4 *>(out + 4 * i) = f32x4{tc_bf16_lo(h[0]) + tc_bf16_lo(l[0]), tc_bf16_hi(h[0]) + tc_bf16_hi(l[0]),
                                                        tc_bf16_lo(h[1]) + tc_bf16_lo(l[1]), tc_bf16_hi(h[1]) + tc_bf16_hi(l[1])};
    }
}

// One wave per output feature n: Wf[n,:] = bf16(gamma o W[n,:]), s[n] = sum_k Wf[n,k] (of the rounded
// values: it must cancel what the product accumulates), c[n] = sum_k beta[k] W[n,k] + bias[n].
__global__ __launch_bounds__(256) void ln_fold_kernel(const float *__restrict__ W, const float *__restrict__ gamma,
                                                      const float *__restrict__ beta, const float *__restrict__ bias,
                                                      uint16_t *__restrict__ Wf, float *__restrict__ s_out,
                                                      float *__restrict__ c_out, int N, int K) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float s = 0.f, c = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
        const f32x4 w = *reinterpret_cast<const f32x4 *>(W + (long)n * K + k);
        const f32x4 g = *reinterpret_cast<const f32x4 *>(gamma + k);
        const f32x4 b = *reinterpret_cast<const f32x4 *>(beta + k);
        const f32x4 f = w * g;
        u32x2 o;
        o[0] = tc_pack_bf16(f[0], f[1]);
        o[1] = tc_pack_bf16(f[2], f[3]);
        *reinterpret_cast<u32x2 *>(Wf + (long)n * K + k) = o;
        s += (tc_bf16_lo(o[0]) + tc_bf16_hi(o[0])) + (tc_bf16_lo(o[1]) + tc_bf16_hi(o[1]));
        c += (w[0] * b[0] + w[1] * b[1]) + (w[2] * b[2] + w[3] * b[3]);
    }
    s = tc_wave_sum(s);
    c = tc_wave_sum(c);
    if (lane == 0) {
        s_out[n] = s;
        c_out[n] = c + (bias ? bias[n] : 0.f);
    }
}

// ---- LayerNorm -----------------------------------------------------------------------
// NCH = ceil(E / 256): each lane keeps NCH float4 of its row in registers, so the row is
// read from HBM once.
template <int NCH>
__global__ __launch_bounds__(256) void layernorm_kernel(const float *__restrict__ x,
                                                        const float *__restrict__ gamma,
                                                        const float *__restrict__ beta,
                                                        float *out_f32, uint16_t *out_bf16, int M,
                                                        int E, float eps, const uint16_t *__restrict__ x_hi,
                                                        const uint16_t *__restrict__ x_lo) {
    // x_hi / x_lo (instead of x): the row as two bf16 planes, x = hi + lo
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float *src = x + (long)row * E;
    f32x4 v[NCH];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = c * 256 + lane * 4;
        if (col >= E) {
            v[c] = f32x4{0, 0, 0, 0};
        } else if (x_hi) {
            const u32x2 h = *reinterpret_cast<const u32x2 *>(x_hi + (long)row * E + col);
            const u32x2 l = *reinterpret_cast<const u32x2 *>(x_lo + (long)row * E + col);
            v[c] = f32x4{tc_bf16_lo(h[0]) + tc_bf16_lo(l[0]), tc_bf16_hi(h[0]) + tc_bf16_hi(l[0]),
                         tc_bf16_lo(h[1]) + tc_bf16_lo(l[1]), tc_bf16_hi(h[1]) + tc_bf16_hi(l[1])};
        } else {
            v[c] = *reinterpret_cast<const f32x4 *>(src + col);
        }
        s += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
    }
    const float mean = tc_wave_sum(s) / (float)E;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = c * 256 + lane * 4;
        if (col < E) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = v[c][j] - mean;
                q += d * d;
            }
        }
    }
    const float rstd = rsqrtf(tc_wave_sum(q) / (float)E + eps);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = c * 256 + lane * 4;
        if (col < E) {
            const f32x4 g = *reinterpret_cast<const f32x4 *>(gamma + col);
            const f32x4 b = *reinterpret_cast<const f32x4 *>(beta + col);
            f32x4 y;
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = (v[c][j] - mean) * rstd * g[j] + b[j];
            if (out_f32) *reinterpret_cast<f32x4 *>(out_f32 + (long)row * E + col) = y;
            if (out_bf16) {
                u32x2 w;
                w[0] = tc_pack_bf16(y[0], y[1]);
                w[1] = tc_pack_bf16(y[2], y[3]);
                *reinterpret_cast<u32x2 *>(out_bf16 + (long)row * E + col) = w;
            }
        }
    }
}

// ---- key-validity words ------------------------------------------------------------------
__global__ __launch_bounds__(64) void key_mask_kernel(const uint8_t *__restrict__ pad_flags,
                                                      uint64_t *__restrict__ words, int T, int nw) {
    const int b = blockIdx.y, w = blockIdx.x, lane = threadIdx.x;
    const int key = w * 64 + lane;
    const bool valid = key < T && pad_flags[(long)b * T + key] == 0;
    const unsigned long long m = __ballot(valid);
    if (lane == 0) words[(long)b * nw + w] = m;
}

// ---- LayerNorm + thin projection heads -------------------------------------------------------
template <typename T, int NCH>
__global__ __launch_bounds__(256) void head_kernel(const T *__restrict__ g, int ldg, int col0,
                                                   const float *__restrict__ gamma,
                                                   const float *__restrict__ beta,
                                                   const float *__restrict__ W,
                                                   const float *__restrict__ bias,
                                                   float *__restrict__ out, int M, int E, int V,
                                                   float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const T *src = g + (long)row * ldg + col0;
    f32x4 v[NCH];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = c * 256 + lane * 4;
        if (col < E) {
            if constexpr (sizeof(T) == 2) {
                const u32x2 w = *reinterpret_cast<const u32x2 *>(src + col);
                v[c] = f32x4{tc_bf16_lo(w[0]), tc_bf16_hi(w[0]), tc_bf16_lo(w[1]), tc_bf16_hi(w[1])};
            } else {
                v[c] = *reinterpret_cast<const f32x4 *>(src + col);
            }
        } else {
            v[c] = f32x4{0, 0, 0, 0};
        }
        s += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
    }
    const float mean = tc_wave_sum(s) / (float)E;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = c * 256 + lane * 4;
        if (col < E) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = v[c][j] - mean;
                q += d * d;
            }
        }
    }
    const float rstd = rsqrtf(tc_wave_sum(q) / (float)E + eps);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = c * 256 + lane * 4;
        if (col < E) {
            const f32x4 gm = *reinterpret_cast<const f32x4 *>(gamma + col);
            const f32x4 bt = *reinterpret_cast<const f32x4 *>(beta + col);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[c][j] = (v[c][j] - mean) * rstd * gm[j] + bt[j];
        }
    }
    float mine = 0.f;                               // lane o keeps logit o
    for (int o = 0; o < V; ++o) {
        const float *w = W + (long)o * E;
        float d = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int col = c * 256 + lane * 4;
            if (col < E) {
                const f32x4 wv = *reinterpret_cast<const f32x4 *>(w + col);
                d += v[c][0] * wv[0] + v[c][1] * wv[1] + v[c][2] * wv[2] + v[c][3] * wv[3];
            }
        }
        d = tc_wave_sum(d);
        if (lane == o) mine = d + bias[o];
    }
    if (lane < V) out[(long)row * V + lane] = mine;
}

// The same head for the two class counts of the model (V = 10 bases / specials, V = 3 structure
// labels) with this lane's slice of W held in registers: the generic kernel re-reads the whole W
// (30 KB at V = 10) through the L1 for every row, which bounds it at about 1 TB/s of input.  A wave
// keeps 4 V NCH weights and walks rows with a grid stride; arithmetic and reduction order are those
// of the generic kernel, so the results are identical.  What is left is the chain of V + 2 wave
// reductions per row (a joint halving exchange of the V sums compiled to worse code and ran slower).
template <typename T, int NCH, int V>
__global__ __launch_bounds__(256) void head_rows_kernel(const T *__restrict__ g, int ldg, int col0,
                                                        const float *__restrict__ gamma,
                                                        const float *__restrict__ beta,
                                                        const float *__restrict__ W,
                                                        const float *__restrict__ bias,
                                                        float *__restrict__ out, int M, int E, float eps) {
    const int lane = threadIdx.x & 63;
    f32x4 wreg[V][NCH], gm[NCH], bt[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = c * 256 + lane * 4;
        const bool ok = col < E;
        gm[c] = ok ? *reinterpret_cast<const f32x4 *>(gamma + col) : f32x4{0, 0, 0, 0};
        bt[c] = ok ? *reinterpret_cast<const f32x4 *>(beta + col) : f32x4{0, 0, 0, 0};
#pragma unroll
        for (int o = 0; o < V; ++o)
            wreg[o][c] = ok ? *reinterpret_cast<const f32x4 *>(W + (long)o * E + col) : f32x4{0, 0, 0, 0};
    }
    const float my_bias = lane < V ? bias[lane] : 0.f;
    // The next row's loads are issued before this row's arithmetic, so their latency is covered
    // by the V + 2 reductions instead of being paid once per row.
    using Raw = typename std::conditional<sizeof(T) == 2, u32x2, f32x4>::type;
    Raw raw[NCH];
    const int stride = gridDim.x * 4;
    int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row < M) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int col = c * 256 + lane * 4;
            if (col < E) raw[c] = *reinterpret_cast<const Raw *>(g + (long)row * ldg + col0 + col);
        }
    }
    for (; row < M; row += stride) {
        f32x4 v[NCH];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int col = c * 256 + lane * 4;
            if (col < E) {
                if constexpr (sizeof(T) == 2) {
                    v[c] = f32x4{tc_bf16_lo(raw[c][0]), tc_bf16_hi(raw[c][0]), tc_bf16_lo(raw[c][1]), tc_bf16_hi(raw[c][1])};
                } else {
                    v[c] = raw[c];
                }
            } else {
                v[c] = f32x4{0, 0, 0, 0};
            }
            s += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
        }
        if (row + stride < M) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int col = c * 256 + lane * 4;
                if (col < E) raw[c] = *reinterpret_cast<const Raw *>(g + (long)(row + stride) * ldg + col0 + col);
            }
        }
        const float mean = tc_wave_sum(s) / (float)E;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int col = c * 256 + lane * 4;
            if (col < E) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = v[c][j] - mean;
                    q += d * d;
                }
            }
        }
        const float rstd = rsqrtf(tc_wave_sum(q) / (float)E + eps);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int col = c * 256 + lane * 4;
            if (col < E) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[c][j] = (v[c][j] - mean) * rstd * gm[c][j] + bt[c][j];
            }
        }
        float mine = 0.f;                           // lane o keeps logit o
#pragma unroll
        for (int o = 0; o < V; ++o) {
            float d = 0.f;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int col = c * 256 + lane * 4;
                if (col < E) {
                    const f32x4 wv = wreg[o][c];
                    d += v[c][0] * wv[0] + v[c][1] * wv[1] + v[c][2] * wv[2] + v[c][3] * wv[3];
                }
            }
            d = tc_wave_sum(d);
            if (lane == o) mine = d + my_bias;
        }
        if (lane < V) out[(long)row * V + lane] = mine;
    }
}

// The same head on the matrix unit, for bf16 input (the head dense layers' GELU output).  With G = gamma o W,
//     logit[o] = rstd (sum_k v[k] G[o][k] - mean s[o]) + c[o],   s[o] = sum_k G[o][k],  c[o] = sum_k beta[k] W[o][k] + bias[o],
// so one pass over a row gives everything: the V products with G, sum v (a row of ones appended to G) and
// sum v^2 (the diagonal of v v^T: the row fragments serve as BOTH operands of a third MFMA, whose A and B layouts
// hold the same registers).  G stays f32-accurate as a bf16 pair hi + lo (two MFMAs into one accumulator).  A
// wave takes 16 rows at a time: lane (row j = lane & 15, k group = lane >> 4) loads 16 bytes per 32-column step,
// 24 loads in flight, then 72 MFMAs; no per-row chain of wave reductions is left (the register-weights kernel
// above is bound by those: 350 vector instructions per row), the kernel runs at the speed of its loads.
template <int V>
__global__ __launch_bounds__(256) void head_mfma_kernel(const uint16_t *__restrict__ g, int ldg, int col0,
                                                        const float *__restrict__ gamma, const float *__restrict__ beta,
                                                        const float *__restrict__ W, const float *__restrict__ bias,
                                                        float *__restrict__ out, int M, int E, float eps) {
    static_assert(V >= 1 && V <= 14, "V products + the row of ones fit one 16-row operand");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int steps = E >> 5;                        // 32 columns per MFMA
    // LDS: fragment (step, plane) of G at ((step * 2 + plane) * 64 + lane) * 16; then s[16], c[16]
    float *sc = reinterpret_cast<float *>(smem + steps * 2 * 1024);
    {
        // Every load of this set-up is a 16-byte load, and a group of them (four k steps, or one output's row of
        // W with gamma and beta) is in flight at once: the set-up costs a handful of memory latencies per
        // workgroup instead of one per k step and column, and stays within the registers of the row loop.
        const int i = lane & 15, kg = lane >> 4;     // operand row (output) i, k group
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {       // E <= 1024: at most 8 steps per wave
            f32x4 gv[4][2], wv[4][2];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int st = wid + 4 * (4 * half + u);
                if (st < steps) {
                    const int k0 = st * 32 + kg * 8;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        gv[u][h] = *reinterpret_cast<const f32x4 *>(gamma + k0 + 4 * h);
                        wv[u][h] = i < V ? *reinterpret_cast<const f32x4 *>(W + (long)i * E + k0 + 4 * h)
                                         : f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int st = wid + 4 * (4 * half + u);
                if (st < steps) {
                    u32x4 hi, lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float a = gv[u][e >> 1][(2 * e) & 3] * wv[u][e >> 1][(2 * e) & 3];
                        float b = gv[u][e >> 1][(2 * e + 1) & 3] * wv[u][e >> 1][(2 * e + 1) & 3];
                        if (i == V) a = b = 1.0f;
                        hi[e] = tc_pack_bf16(a, b);
                        lo[e] = tc_pack_bf16(a - tc_bf16_lo(hi[e]), b - tc_bf16_hi(hi[e]));
                    }
                    *reinterpret_cast<u32x4 *>(smem + ((st * 2 + 0) * 64 + lane) * 16) = hi;
                    *reinterpret_cast<u32x4 *>(smem + ((st * 2 + 1) * 64 + lane) * 16) = lo;
                }
            }
        }
        // s and c of output o by wave o & 3 (sums of the values the MFMAs multiply: hi + lo)
#pragma unroll 1
        for (int o = wid; o < V; o += 4) {
            f32x4 sg[4], sb[4], sw[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int k = (c * 64 + lane) * 4;
                if (k < E) {
                    sg[c] = *reinterpret_cast<const f32x4 *>(gamma + k);
                    sb[c] = *reinterpret_cast<const f32x4 *>(beta + k);
                    sw[c] = *reinterpret_cast<const f32x4 *>(W + (long)o * E + k);
                }
            }
            float s1 = 0.f, c1 = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if ((c * 64 + lane) * 4 < E) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float gw = sg[c][e] * sw[c][e];
                        const uint32_t h = tc_pack_bf16(gw, 0.f);
                        const uint32_t l = tc_pack_bf16(gw - tc_bf16_lo(h), 0.f);
                        s1 += tc_bf16_lo(h) + tc_bf16_lo(l);
                        c1 += sb[c][e] * sw[c][e];
                    }
                }
            }
            s1 = tc_wave_sum(s1);
            c1 = tc_wave_sum(c1);
            if (lane == 0) { sc[o] = s1; sc[16 + o] = c1 + bias[o]; }
        }
    }
    __syncthreads();
    const int j = lane & 15, rg = lane >> 4;
    const float inv_e = 1.0f / (float)E;
    const int nblk = (M + 15) >> 4;
    for (int blk = blockIdx.x * 4 + wid; blk < nblk; blk += gridDim.x * 4) {
        const int row = blk * 16 + j;
        const uint16_t *src = g + (long)(row < M ? row : M - 1) * ldg + col0 + rg * 8;
        u32x4 vf[32];                                // E <= 1024
#pragma unroll
        for (int st = 0; st < 32; ++st)
            if (st < steps) vf[st] = *reinterpret_cast<const u32x4 *>(src + st * 32);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, gram = acc;
#pragma unroll
        for (int st = 0; st < 32; ++st) {
            if (st < steps) {
                const bf16x8 v = __builtin_bit_cast(bf16x8, vf[st]);
                const bf16x8 gh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(smem + ((st * 2 + 0) * 64 + lane) * 16));
                const bf16x8 gl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(smem + ((st * 2 + 1) * 64 + lane) * 16));
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gh, v, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gl, v, acc, 0, 0, 0);
                gram = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v, v, gram, 0, 0, 0);
            }
        }
        // lane (j, rg) holds acc[e] = D[o = 4 rg + e][row j] and gram[e] = (v v^T)[4 rg + e][j]
        const float own_sum = acc[V & 3];            // meaningful in the lanes with rg == V >> 2
        const float diag = (j & 3) == 0 ? gram[0] : (j & 3) == 1 ? gram[1] : (j & 3) == 2 ? gram[2] : gram[3];
        const float sum_v = __shfl(own_sum, j + 16 * (V >> 2), 64);
        const float sum_v2 = __shfl(diag, j + 16 * (j >> 2), 64);
        const float mean = sum_v * inv_e;
        const float var = fmaxf(sum_v2 * inv_e - mean * mean, 0.f);
        const float rstd = rsqrtf(var + eps);
        if (row < M) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int o = rg * 4 + e;
                if (o < V) out[(long)row * V + o] = rstd * (acc[e] - mean * sc[o]) + sc[16 + o];
            }
        }
    }
}

// ---- greedy synonymous-codon decode ----------------------------------------------------------
// Amino-acid code = index into "FLIMVPTAY_HQNKDECWRSG"; codons in the table order of the
// reference; bases coded 0..3 = A,U,C,G (logit columns 5..8).
__constant__ uint8_t kNumCodons[21] = {2, 6, 3, 1, 4, 4, 4, 4, 2, 3, 2, 2, 2, 2, 2, 2, 2, 1, 6, 6, 4};
#define CD(a, b, c) (uint8_t)((a) | ((b) << 2) | ((c) << 4))
enum { bA = 0, bU = 1, bC = 2, bG = 3 };
__constant__ uint8_t kCodons[21][6] = {
    /*F*/ {CD(bU, bU, bU), CD(bU, bU, bC)},
    /*L*/ {CD(bU, bU, bA), CD(bU, bU, bG), CD(bC, bU, bU), CD(bC, bU, bC), CD(bC, bU, bA), CD(bC, bU, bG)},
    /*I*/ {CD(bA, bU, bU), CD(bA, bU, bC), CD(bA, bU, bA)},
    /*M*/ {CD(bA, bU, bG)},
    /*V*/ {CD(bG, bU, bU), CD(bG, bU, bC), CD(bG, bU, bA), CD(bG, bU, bG)},
    /*P*/ {CD(bC, bC, bU), CD(bC, bC, bC), CD(bC, bC, bA), CD(bC, bC, bG)},
    /*T*/ {CD(bA, bC, bU), CD(bA, bC, bC), CD(bA, bC, bA), CD(bA, bC, bG)},
    /*A*/ {CD(bG, bC, bU), CD(bG, bC, bC), CD(bG, bC, bA), CD(bG, bC, bG)},
    /*Y*/ {CD(bU, bA, bU), CD(bU, bA, bC)},
    /*_*/ {CD(bU, bA, bA), CD(bU, bA, bG), CD(bU, bG, bA)},
    /*H*/ {CD(bC, bA, bU), CD(bC, bA, bC)},
    /*Q*/ {CD(bC, bA, bA), CD(bC, bA, bG)},
    /*N*/ {CD(bA, bA, bU), CD(bA, bA, bC)},
    /*K*/ {CD(bA, bA, bA), CD(bA, bA, bG)},
    /*D*/ {CD(bG, bA, bU), CD(bG, bA, bC)},
    /*E*/ {CD(bG, bA, bA), CD(bG, bA, bG)},
    /*C*/ {CD(bU, bG, bU), CD(bU, bG, bC)},
    /*W*/ {CD(bU, bG, bG)},
    /*R*/ {CD(bC, bG, bU), CD(bC, bG, bC), CD(bC, bG, bA), CD(bC, bG, bG), CD(bA, bG, bA), CD(bA, bG, bG)},
    /*S*/ {CD(bA, bG, bU), CD(bA, bG, bC), CD(bU, bC, bU), CD(bU, bC, bC), CD(bU, bC, bA), CD(bU, bC, bG)},
    /*G*/ {CD(bG, bG, bU), CD(bG, bG, bC), CD(bG, bG, bA), CD(bG, bG, bG)},
};

__global__ __launch_bounds__(256) void decode_kernel(const float *__restrict__ logits, int ldl,
                                                     const uint8_t *__restrict__ aa,
                                                     const int32_t *__restrict__ aa_off,
                                                     const int32_t *__restrict__ row0,
                                                     uint8_t *__restrict__ bases, int n_seq,
                                                     int n_aa_total) {
    const int i = blockIdx.x * 256 + threadIdx.x;       // global residue index
    if (i >= n_aa_total) return;
    // sequence of residue i: aa_off is sorted, binary search for the last offset <= i
    int lo = 0, hi = n_seq;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (aa_off[mid] <= i) lo = mid; else hi = mid;
    }
    const int local = i - aa_off[lo];
    const long row = (long)row0[lo] + 3L * local;
    float pb[3][4];                                      // softmax probability of A,U,C,G per position
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float *l = logits + (row + j) * ldl;
        float v[10];
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < 10; ++c) { v[c] = l[c]; mx = fmaxf(mx, v[c]); }
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < 10; ++c) {
            // exp in f64 then one rounding to f32: keeps subnormal probabilities (the f32
            // library exp flushes them), which decide ties between underflowing codons
            v[c] = (float)exp((double)(v[c] - mx));
            sum += v[c];
        }
        const float inv = 1.0f / sum;
#pragma unroll
        for (int c = 0; c < 4; ++c) pb[j][c] = v[5 + c] * inv;
    }
    const int a = aa[i];
    const int nc = a < 21 ? kNumCodons[a] : 0;
    double best = -INFINITY;
    uint8_t pick = 0;
    for (int c = 0; c < nc; ++c) {
        const uint8_t cd = kCodons[a][c];
        double pr = 1.0;
        pr *= (double)pb[0][cd & 3];
        pr *= (double)pb[1][(cd >> 2) & 3];
        pr *= (double)pb[2][(cd >> 4) & 3];
        if (pr > best) { best = pr; pick = cd; }
    }
    const char sym[4] = {'A', 'U', 'C', 'G'};
    uint8_t *o = bases + 3L * i;
    if (nc == 0) { o[0] = o[1] = o[2] = '?'; return; }
    o[0] = sym[pick & 3];
    o[1] = sym[(pick >> 2) & 3];
    o[2] = sym[(pick >> 4) & 3];
}

// ---- cross-entropy over thin logits --------------------------------------------------------
__global__ __launch_bounds__(256) void ce_kernel(const float *__restrict__ logits,
                                                 const int32_t *__restrict__ labels,
                                                 float *loss_sum, int32_t *count, float *dlogits,
                                                 float grad_scale, int M, int V) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    float loss = 0.f;
    int cnt = 0;
    if (row < M) {
        const int y = labels[row];
        const float *l = logits + (long)row * V;
        if (y >= 0 && y < V) {
            float mx = -INFINITY;
            for (int c = 0; c < V; ++c) mx = fmaxf(mx, l[c]);
            float sum = 0.f;
            for (int c = 0; c < V; ++c) sum += expf(l[c] - mx);
            const float lse = mx + logf(sum);
            loss = lse - l[y];
            cnt = 1;
            if (dlogits) {
                for (int c = 0; c < V; ++c)
                    dlogits[(long)row * V + c] = (expf(l[c] - lse) - (c == y ? 1.f : 0.f)) * grad_scale;
            }
        } else {
            // Only -100 is "ignore" (torch.nn.CrossEntropyLoss raises on any other label outside
            // [0, V)); a kernel cannot raise, so such a label poisons the loss with NaN instead of
            // silently shrinking the denominator.
            if (y != -100) loss = __builtin_nanf("");
            if (dlogits)
                for (int c = 0; c < V; ++c) dlogits[(long)row * V + c] = 0.f;
        }
    }
    const bool poisoned = __ballot(loss != loss) != 0ull;
    if (poisoned && (threadIdx.x & 63) == 0) atomicAdd(loss_sum, __builtin_nanf(""));
    loss = tc_wave_sum(loss);
    const unsigned long long voted = __ballot(cnt != 0);
    if ((threadIdx.x & 63) == 0) {
        const int n = __popcll(voted);
        if (n) {
            atomicAdd(loss_sum, loss);
            atomicAdd(count, n);
        }
    }
}

// ---- masked mean pooling -------------------------------------------------------------------
// grid (ceil(E/256), B): 64 row-lanes x 4 columns per thread... one thread per column, rows looped
__global__ __launch_bounds__(256) void mean_pool_kernel(const float *__restrict__ rep,
                                                        const int32_t *__restrict__ start,
                                                        const int32_t *__restrict__ len,
                                                        float *__restrict__ pooled, int T, int E,
                                                        int packed_rows) {
    // packed_rows > 0: start[] holds absolute rows of a packed buffer of that many rows
    const int b = blockIdx.y;
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= E) return;
    int s0 = start[b], n = len[b];
    s0 = s0 < 0 ? 0 : s0;
    const int limit = packed_rows > 0 ? packed_rows : T;
    if (s0 + n > limit) n = limit - s0;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const float *base = rep + ((packed_rows > 0 ? 0L : (long)b * T) + s0) * E + col;
    int t = 0;
    for (; t + 4 <= n; t += 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += base[(long)(t + j) * E];
    }
    for (; t < n; ++t) acc[0] += base[(long)t * E];
    pooled[(long)b * E + col] = n > 0 ? ((acc[0] + acc[1]) + (acc[2] + acc[3])) / (float)n : 0.f;
}

// ---- per-sequence codon statistics of decoded bases ----------------------------------------------
// One workgroup per sequence over its codons [off[s], off[s+1]); integer counts only, so the ratios
// the caller forms equal those of a host loop exactly.
__global__ __launch_bounds__(256) void codon_stats_kernel(const uint8_t *__restrict__ bases,
                                                          const uint8_t *__restrict__ ref,
                                                          const int32_t *__restrict__ off,
                                                          const uint8_t *__restrict__ flag64,
                                                          int32_t *__restrict__ out) {
    __shared__ int acc[4];
    if (threadIdx.x < 4) acc[threadIdx.x] = 0;
    __syncthreads();
    const int s = blockIdx.x;
    const int first = off[s], last = off[s + 1];
    int gc = 0, match = 0, flagged = 0, bad = 0;
    for (int i = first + threadIdx.x; i < last; i += 256) {
        int idx = 0;
        bool same = ref != nullptr;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const uint8_t c = bases[3L * i + j];
            const int v = c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : (c == 'T' || c == 'U') ? 3 : -1;
            gc += (c == 'G' || c == 'C');
            if (v < 0) bad = 1;
            idx = idx * 4 + (v < 0 ? 0 : v);
            if (ref) same = same && ref[3L * i + j] == c;
        }
        match += same;
        if (flag64) flagged += flag64[idx];
    }
    gc = (int)tc_wave_sum((float)gc);
    match = (int)tc_wave_sum((float)match);
    flagged = (int)tc_wave_sum((float)flagged);
    bad = __ballot(bad != 0) != 0ull;
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&acc[0], gc);
        atomicAdd(&acc[1], match);
        atomicAdd(&acc[2], flagged);
        atomicOr(&acc[3], bad);
    }
    __syncthreads();
    if (threadIdx.x < 4) out[4 * s + threadIdx.x] = acc[threadIdx.x];
}

// ---- %MinMax windows of decoded bases -------------------------------------------------------------
// One thread per codon position.  Position i of a sequence of n codons with i + window <= n sums the four
// table values of codons i .. i + window - 1 in that order in f64, as the host loop does, so the value
// equals the host's to the last bit; every other position (the tail a window does not fit) gets NaN.
// A letter outside ACGTU, or a codon whose table entry is NaN, sets *status to 1 (and the windows that cover it
// hold NaN); a window whose amino acids all have a single codon divides by zero and holds what IEEE gives.
__global__ __launch_bounds__(256) void codon_minmax_kernel(const uint8_t *__restrict__ bases,
                                                           const int32_t *__restrict__ off,
                                                           const double *__restrict__ tab, int window, int n_seq,
                                                           long n_codons, double *__restrict__ out,
                                                           int32_t *__restrict__ status) {
    __shared__ double t[4 * 64];
    for (int e = threadIdx.x; e < 256; e += 256) t[e] = tab[e];
    __syncthreads();
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_codons) return;
    // the sequence that owns codon i: last s with off[s] <= i
    int lo = 0, hi = n_seq;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if ((long)off[mid] <= i) lo = mid; else hi = mid;
    }
    const long last = off[lo + 1];
    const double nan = __builtin_nan("");
    if (i + window > last) { out[i] = nan; return; }
    double actual = 0.0, mx = 0.0, mn = 0.0, avg = 0.0;
    bool bad = false;
    for (int w = 0; w < window; ++w) {
        int idx = 0;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const uint8_t c = bases[3L * (i + w) + j];
            const int v = c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : (c == 'T' || c == 'U') ? 3 : -1;
            bad |= v < 0;
            idx = idx * 4 + (v < 0 ? 0 : v);
        }
        bad |= t[idx] != t[idx];
        actual += t[idx];
        mx += t[64 + idx];
        mn += t[128 + idx];
        avg += t[192 + idx];
    }
    const double ws = (double)window;
    actual /= ws; mx /= ws; mn /= ws; avg /= ws;
    const double pmax = ((actual - avg) / (mx - avg)) * 100.0;
    const double pmin = ((avg - actual) / (avg - mn)) * 100.0;
    out[i] = bad ? nan : (pmax >= 0.0 ? pmax : -pmin);
    if (bad) atomicOr(status, 1);
}

}  // namespace

extern "C" int tc_codon_minmax(const uint8_t *bases, const int32_t *codon_off, const double *tables, int window,
                               double *out, int32_t *status, int n_seq, long n_codons, void *stream) {
    TC_REQUIRE(bases && codon_off && tables && out && status && n_seq > 0 && n_codons > 0 && window > 0,
               "tc_codon_minmax: bad arguments");
    TC_LAUNCH(codon_minmax_kernel, dim3((unsigned)((n_codons + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
              bases, codon_off, tables, window, n_seq, n_codons, out, status);
    TC_CHECK_LAUNCH("tc_codon_minmax");
    return TC_OK;
}

extern "C" int tc_codon_stats(const uint8_t *bases, const uint8_t *ref_bases, const int32_t *codon_off,
                              const uint8_t *codon_flags, int32_t *counts, int n_seq, void *stream) {
    TC_REQUIRE(bases && codon_off && counts && n_seq > 0, "tc_codon_stats: bad arguments");
    TC_LAUNCH(codon_stats_kernel, dim3(n_seq), dim3(256), 0, static_cast<hipStream_t>(stream), bases, ref_bases,
              codon_off, codon_flags, counts);
    TC_CHECK_LAUNCH("tc_codon_stats");
    return TC_OK;
}

extern "C" int tc_mean_pool(const float *rep, const int32_t *start, const int32_t *len, float *pooled,
                            int B, int T, int E, void *stream) {
    TC_REQUIRE(rep && start && len && pooled, "tc_mean_pool: null operand");
    TC_REQUIRE(B > 0 && T > 0 && E > 0 && B <= 65535, "tc_mean_pool: bad shape B=%d T=%d E=%d", B, T, E);
    TC_LAUNCH(mean_pool_kernel, dim3((E + 255) / 256, B), dim3(256), 0,
                       static_cast<hipStream_t>(stream), rep, start, len, pooled, T, E, 0);
    TC_CHECK_LAUNCH("tc_mean_pool");
    return TC_OK;
}

extern "C" int tc_mean_pool_packed(const float *rep, const int32_t *row_start, const int32_t *len, float *pooled,
                                   int n_seq, int total_rows, int E, void *stream) {
    TC_REQUIRE(rep && row_start && len && pooled, "tc_mean_pool_packed: null operand");
    TC_REQUIRE(n_seq > 0 && total_rows > 0 && E > 0, "tc_mean_pool_packed: bad shape n_seq=%d rows=%d E=%d", n_seq,
               total_rows, E);
    for (int first = 0; first < n_seq; first += 65535) {          // grid.y limit
        const int n = n_seq - first < 65535 ? n_seq - first : 65535;
        TC_LAUNCH(mean_pool_kernel, dim3((E + 255) / 256, n), dim3(256), 0, static_cast<hipStream_t>(stream), rep,
                  row_start + first, len + first, pooled + (long)first * E, 0, E, total_rows);
        TC_CHECK_LAUNCH("tc_mean_pool_packed");
    }
    return TC_OK;
}

static int embed_impl(const int32_t *tokens, const int32_t *species, const float *tok_table,
                      const float *species_table, float *x, uint8_t *pad_flags, int32_t *bad_index, int B, int T,
                      int E, int vocab, int n_species, int pad_idx, void *xb, float *ln_rows, float eps,
                      void *stream, void *x_lo = nullptr) {
    TC_REQUIRE(tokens && species && tok_table && species_table && (x || (xb && x_lo)), "tc_embed_fwd: null operand");
    TC_REQUIRE(tc_aligned16(x_lo) && (!x_lo || xb), "tc_embed_fwd_ln: the remainder plane comes with the bf16 plane, 16-byte aligned");
    TC_REQUIRE(B > 0 && T > 0 && E > 0 && E % 4 == 0, "tc_embed_fwd: bad shape B=%d T=%d E=%d", B, T, E);
    TC_REQUIRE(tc_aligned16(tok_table) && tc_aligned16(species_table) && tc_aligned16(x),
               "tc_embed_fwd: tables and output must be 16-byte aligned");
    TC_REQUIRE((xb == nullptr) == (ln_rows == nullptr) && tc_aligned16(xb) && tc_aligned16(ln_rows),
               "tc_embed_fwd_ln: the bf16 copy and the row statistics come together, 16-byte aligned");
    const long M = (long)B * T;
    TC_REQUIRE(M < (1L << 31), "tc_embed_fwd: too many rows");
    hipStream_t s = static_cast<hipStream_t>(stream);
    TcProfScope prof(TC_PROF_EMBED, s, (double)M * (4.0 + (xb ? 6.0 : 4.0) * E));
    TC_LAUNCH(embed_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, tokens, species,
                       tok_table, species_table, x, pad_flags, bad_index, (int)M, T, E, vocab,
                       n_species, pad_idx, static_cast<uint16_t *>(xb), ln_rows, eps, static_cast<uint16_t *>(x_lo));
    TC_CHECK_LAUNCH("tc_embed_fwd");
    return TC_OK;
}

extern "C" int tc_embed_fwd(const int32_t *tokens, const int32_t *species, const float *tok_table,
                            const float *species_table, float *x, uint8_t *pad_flags,
                            int32_t *bad_index, int B, int T, int E, int vocab, int n_species,
                            int pad_idx, void *stream) {
    return embed_impl(tokens, species, tok_table, species_table, x, pad_flags, bad_index, B, T, E, vocab, n_species,
                      pad_idx, nullptr, nullptr, 0.f, stream);
}

extern "C" int tc_embed_fwd_ln(const int32_t *tokens, const int32_t *species, const float *tok_table,
                               const float *species_table, float *x, uint8_t *pad_flags, int32_t *bad_index, int B,
                               int T, int E, int vocab, int n_species, int pad_idx, void *xb, void *x_lo,
                               float *ln_rows, float eps, void *stream) {
    TC_REQUIRE(xb && ln_rows, "tc_embed_fwd_ln: null operand");
    return embed_impl(tokens, species, tok_table, species_table, x, pad_flags, bad_index, B, T, E, vocab, n_species,
                      pad_idx, xb, ln_rows, eps, stream, x_lo);
}

extern "C" int tc_ln_rows_finalize(unsigned long long *row_acc, float *ln_rows, int M, int E, float eps, void *stream) {
    TC_REQUIRE(row_acc && ln_rows && M > 0 && E > 0, "tc_ln_rows_finalize: bad operand");
    hipStream_t s = static_cast<hipStream_t>(stream);
    TcProfScope prof(TC_PROF_LN_ROWS, s, (double)M * 40.0);        // 16 B read, 16 B cleared, 8 B written per row
    TC_LAUNCH(ln_rows_finalize_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, row_acc, ln_rows, M, E, eps);
    TC_CHECK_LAUNCH("tc_ln_rows_finalize");
    return TC_OK;
}

extern "C" int tc_splitk_bias_residual(const float *slabs, int S, const float *bias, float *x, int M, int N, void *stream) {
    TC_REQUIRE(slabs && bias && x && S >= 1 && M > 0 && N > 0 && N % 4 == 0, "tc_splitk_bias_residual: bad operand");
    TC_REQUIRE(tc_aligned16(slabs) && tc_aligned16(bias) && tc_aligned16(x), "tc_splitk_bias_residual: operands must be 16-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long n4 = (long)M * N / 4;
    const unsigned grid = (unsigned)((n4 + 255) / 256 < 65536 ? (n4 + 255) / 256 : 65536);
    TC_LAUNCH(splitk_bias_residual_kernel, dim3(grid), dim3(256), 0, s, slabs, S, (long)M * N, bias, x, n4, N);
    TC_CHECK_LAUNCH("tc_splitk_bias_residual");
    return TC_OK;
}

extern "C" int tc_copy_f32(float *dst, const float *src, long n, void *stream) {
    TC_REQUIRE(dst && src && n > 0 && n % 4 == 0 && tc_aligned16(dst) && tc_aligned16(src), "tc_copy_f32: bad operand");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long n4 = n / 4;
    const unsigned grid = (unsigned)((n4 + 255) / 256 < 65536 ? (n4 + 255) / 256 : 65536);
    TC_LAUNCH(copy_f32_kernel, dim3(grid), dim3(256), 0, s, src, dst, n4);
    TC_CHECK_LAUNCH("tc_copy_f32");
    return TC_OK;
}

extern "C" int tc_ln_rows_clear(unsigned long long *row_acc, int M, void *stream) {
    TC_REQUIRE(row_acc && M > 0 && tc_aligned16(row_acc), "tc_ln_rows_clear: bad operand");
    hipStream_t s = static_cast<hipStream_t>(stream);
    TC_LAUNCH(ln_rows_clear_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, row_acc, (long)M);
    TC_CHECK_LAUNCH("tc_ln_rows_clear");
    return TC_OK;
}

extern "C" int tc_ln_fold_weights(const float *W, const float *gamma, const float *beta, const float *bias, void *Wf,
                                  float *s_out, float *c_out, int N, int K, void *stream) {
    TC_REQUIRE(W && gamma && beta && Wf && s_out && c_out, "tc_ln_fold_weights: null operand");
    TC_REQUIRE(N > 0 && K > 0 && K % 4 == 0, "tc_ln_fold_weights: bad shape N=%d K=%d", N, K);
    TC_REQUIRE(tc_aligned16(W) && tc_aligned16(gamma) && tc_aligned16(beta) && tc_aligned16(Wf),
               "tc_ln_fold_weights: operands must be 16-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    TC_LAUNCH(ln_fold_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, s, W, gamma, beta, bias,
              static_cast<uint16_t *>(Wf), s_out, c_out, N, K);
    TC_CHECK_LAUNCH("tc_ln_fold_weights");
    return TC_OK;
}

static int layernorm_impl(const float *x, const uint16_t *x_hi, const uint16_t *x_lo, const float *gamma,
                          const float *beta, float *out_f32, void *out_bf16, int M, int E, float eps, void *stream) {
    TC_REQUIRE((x || (x_hi && x_lo)) && gamma && beta && (out_f32 || out_bf16), "tc_layernorm_fwd: null operand");
    TC_REQUIRE(M > 0 && E > 0 && E % 4 == 0 && E <= 2048, "tc_layernorm_fwd: bad shape M=%d E=%d", M, E);
    TC_REQUIRE(tc_aligned16(x) && tc_aligned16(x_hi) && tc_aligned16(x_lo) && tc_aligned16(gamma) && tc_aligned16(beta) &&
                   tc_aligned16(out_f32) && (((uintptr_t)out_bf16) & 7u) == 0,
               "tc_layernorm_fwd: misaligned operand");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const double bytes = (double)M * E * (4.0 + (out_f32 ? 4.0 : 0.0) + (out_bf16 ? 2.0 : 0.0));
    TcProfScope prof(TC_PROF_LAYERNORM, s, bytes);
    const dim3 grid((M + 3) / 4), block(256);
    uint16_t *ob = static_cast<uint16_t *>(out_bf16);
    const int nch = (E + 255) / 256;
#define TC_LN(N) TC_LAUNCH(layernorm_kernel<N>, grid, block, 0, s, x, gamma, beta, out_f32, ob, M, E, eps, x_hi, x_lo)
    switch (nch) {
        case 1: TC_LN(1); break;
        case 2: TC_LN(2); break;
        case 3: TC_LN(3); break;
        case 4: TC_LN(4); break;
        default: TC_LN(8); break;
    }
#undef TC_LN
    TC_CHECK_LAUNCH("tc_layernorm_fwd");
    return TC_OK;
}

extern "C" int tc_layernorm_fwd(const float *x, const float *gamma, const float *beta,
                                float *out_f32, void *out_bf16, int M, int E, float eps,
                                void *stream) {
    TC_REQUIRE(x, "tc_layernorm_fwd: null operand");
    return layernorm_impl(x, nullptr, nullptr, gamma, beta, out_f32, out_bf16, M, E, eps, stream);
}

extern "C" int tc_layernorm_fwd_split(const void *x_hi, const void *x_lo, const float *gamma, const float *beta,
                                      float *out_f32, void *out_bf16, int M, int E, float eps, void *stream) {
    TC_REQUIRE(x_hi && x_lo, "tc_layernorm_fwd_split: null operand");
    return layernorm_impl(nullptr, static_cast<const uint16_t *>(x_hi), static_cast<const uint16_t *>(x_lo), gamma, beta,
                          out_f32, out_bf16, M, E, eps, stream);
}

extern "C" int tc_split_to_f32(const void *x_hi, const void *x_lo, float *out, long n, void *stream) {
    TC_REQUIRE(x_hi && x_lo && out && n > 0 && n % 4 == 0, "tc_split_to_f32: bad operand");
    TC_REQUIRE(tc_aligned16(x_hi) && tc_aligned16(x_lo) && tc_aligned16(out), "tc_split_to_f32: misaligned operand");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long n4 = n / 4;
    const unsigned grid = (unsigned)((n4 + 255) / 256 < 65536 ? (n4 + 255) / 256 : 65536);
    TC_LAUNCH(split_to_f32_kernel, dim3(grid), dim3(256), 0, s, static_cast<const uint16_t *>(x_hi),
              static_cast<const uint16_t *>(x_lo), out, n4);
    TC_CHECK_LAUNCH("tc_split_to_f32");
    return TC_OK;
}

extern "C" int tc_key_mask_words(const uint8_t *pad_flags, uint64_t *words, int B, int T,
                                 void *stream) {
    TC_REQUIRE(pad_flags && words && B > 0 && T > 0, "tc_key_mask_words: bad arguments");
    const int nw = (T + 63) / 64;
    TC_LAUNCH(key_mask_kernel, dim3(nw, B), dim3(64), 0, static_cast<hipStream_t>(stream),
                       pad_flags, words, T, nw);
    TC_CHECK_LAUNCH("tc_key_mask_words");
    return TC_OK;
}

extern "C" int tc_head_logits_fwd(int dtype, const void *g, int ldg, int col0, const float *gamma,
                                  const float *beta, const float *W, const float *bias, float *out,
                                  int M, int E, int V, float eps, void *stream) {
    TC_REQUIRE(dtype == TC_BF16 || dtype == TC_F32, "tc_head_logits_fwd: bad dtype %d", dtype);
    TC_REQUIRE(g && gamma && beta && W && bias && out, "tc_head_logits_fwd: null operand");
    TC_REQUIRE(M > 0 && E > 0 && E % 4 == 0 && E <= 1024 && V > 0 && V <= 16,
               "tc_head_logits_fwd: bad shape M=%d E=%d V=%d", M, E, V);
    TC_REQUIRE(ldg % 4 == 0 && col0 % 4 == 0 && col0 + E <= ldg, "tc_head_logits_fwd: bad ldg/col0");
    TC_REQUIRE(tc_aligned16(g) && tc_aligned16(gamma) && tc_aligned16(beta) && tc_aligned16(W),
               "tc_head_logits_fwd: misaligned operand");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int esz = dtype == TC_BF16 ? 2 : 4;
    TcProfScope prof(TC_PROF_HEAD, s, (double)M * (E * (double)esz + 4.0 * V));
    const dim3 grid((M + 3) / 4), block(256);
    const int nch = (E + 255) / 256;
    // bf16 rows of the model's widths: the matrix-unit form (TC_HEAD_MFMA=0 keeps the wave-per-row kernels)
    {
        static int mfma = -1;
        if (mfma < 0) { const char *e = getenv("TC_HEAD_MFMA"); mfma = e ? atoi(e) != 0 : 1; }
        if (mfma && dtype == TC_BF16 && (V == 10 || V == 3) && E % 32 == 0 && ldg % 8 == 0 && col0 % 8 == 0 && M >= 16 &&
            tc_aligned16(g) && tc_aligned16(gamma) && tc_aligned16(beta) && tc_aligned16(W)) {
            const int lds = (E / 32) * 2 * 1024 + 128;
            const int nblk = (M + 15) / 16;
            const dim3 mgrid((nblk + 3) / 4 < 768 ? (nblk + 3) / 4 : 768);
            static bool attr10 = false, attr3 = false;
            bool &attr = V == 10 ? attr10 : attr3;
            if (!attr) {
                const void *fn = V == 10 ? reinterpret_cast<const void *>(head_mfma_kernel<10>)
                                         : reinterpret_cast<const void *>(head_mfma_kernel<3>);
                if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024 + 128) != hipSuccess) {
                    tc_set_error("tc_head_logits_fwd: hipFuncSetAttribute failed");
                    return TC_ELAUNCH;
                }
                attr = true;
            }
            if (V == 10)
                TC_LAUNCH((head_mfma_kernel<10>), mgrid, block, lds, s, static_cast<const uint16_t *>(g), ldg, col0, gamma, beta, W,
                          bias, out, M, E, eps);
            else
                TC_LAUNCH((head_mfma_kernel<3>), mgrid, block, lds, s, static_cast<const uint16_t *>(g), ldg, col0, gamma, beta, W,
                          bias, out, M, E, eps);
            TC_CHECK_LAUNCH("tc_head_logits_fwd");
            return TC_OK;
        }
    }
    if (nch == 3 && (V == 10 || V == 3)) {
        // the model's own heads: weights in registers, rows by grid stride (1024 workgroups: 4 per CU)
        const dim3 rgrid((M + 3) / 4 < 1024 ? (M + 3) / 4 : 1024);
#define TC_HEAD_ROWS(TY, VV)                                                                              \
    TC_LAUNCH((head_rows_kernel<TY, 3, VV>), rgrid, block, 0, s, static_cast<const TY *>(g), ldg, col0, \
              gamma, beta, W, bias, out, M, E, eps)
        if (dtype == TC_BF16) { if (V == 10) TC_HEAD_ROWS(uint16_t, 10); else TC_HEAD_ROWS(uint16_t, 3); }
        else { if (V == 10) TC_HEAD_ROWS(float, 10); else TC_HEAD_ROWS(float, 3); }
#undef TC_HEAD_ROWS
        TC_CHECK_LAUNCH("tc_head_logits_fwd");
        return TC_OK;
    }
#define TC_HEAD(TY, N)                                                                          \
    TC_LAUNCH((head_kernel<TY, N>), grid, block, 0, s, static_cast<const TY *>(g), ldg, \
                       col0, gamma, beta, W, bias, out, M, E, V, eps)
    if (dtype == TC_BF16) {
        switch (nch) {
            case 1: TC_HEAD(uint16_t, 1); break;
            case 2: TC_HEAD(uint16_t, 2); break;
            case 3: TC_HEAD(uint16_t, 3); break;
            default: TC_HEAD(uint16_t, 4); break;
        }
    } else {
        switch (nch) {
            case 1: TC_HEAD(float, 1); break;
            case 2: TC_HEAD(float, 2); break;
            case 3: TC_HEAD(float, 3); break;
            default: TC_HEAD(float, 4); break;
        }
    }
#undef TC_HEAD
    TC_CHECK_LAUNCH("tc_head_logits_fwd");
    return TC_OK;
}

extern "C" int tc_codon_decode(const float *logits, int ldl, const uint8_t *aa,
                               const int32_t *aa_off, const int32_t *row0, uint8_t *bases,
                               int n_seq, int n_aa_total, void *stream) {
    TC_REQUIRE(logits && aa && aa_off && row0 && bases, "tc_codon_decode: null operand");
    TC_REQUIRE(ldl >= 10 && n_seq > 0 && n_aa_total >= 0, "tc_codon_decode: bad shape");
    if (n_aa_total == 0) return TC_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    TcProfScope prof(TC_PROF_DECODE, s, (double)n_aa_total * (120.0 + 3.0 + 1.0));
    TC_LAUNCH(decode_kernel, dim3((n_aa_total + 255) / 256), dim3(256), 0, s, logits, ldl, aa,
                       aa_off, row0, bases, n_seq, n_aa_total);
    TC_CHECK_LAUNCH("tc_codon_decode");
    return TC_OK;
}

extern "C" int tc_ce_loss_fwd_bwd(const float *logits, const int32_t *labels, float *loss_sum,
                                  int32_t *count, float *dlogits, float grad_scale, int M, int V,
                                  void *stream) {
    TC_REQUIRE(logits && labels && loss_sum && count, "tc_ce_loss_fwd_bwd: null operand");
    TC_REQUIRE(M > 0 && V > 0 && V <= 64, "tc_ce_loss_fwd_bwd: bad shape M=%d V=%d", M, V);
    TC_LAUNCH(ce_kernel, dim3((M + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                       logits, labels, loss_sum, count, dlogits, grad_scale, M, V);
    TC_CHECK_LAUNCH("tc_ce_loss_fwd_bwd");
    return TC_OK;
}
