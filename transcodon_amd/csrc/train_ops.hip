// Backward-pass row kernels of the training step: transposes for the weight-gradient GEMMs,
// split-K slab reduction, LayerNorm / GELU / thin-head / embedding / rotary backward.
// All of them are HBM-bound; column reductions (bias, LayerNorm affine, thin-head weight
// gradients) are accumulated per wave in registers, combined per workgroup through LDS and
// added to the f32 gradient buffers with one atomic per column and workgroup.
#include "common.h"

namespace {

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<uint16_t>(uint16_t v) {
    return __builtin_bit_cast(float, (uint32_t)v << 16);
}
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ uint16_t from_f32<uint16_t>(float v) {
    return __builtin_bit_cast(uint16_t, (__bf16)v);
}

// ---- transpose + cast (+ column sums) ------------------------------------------------------
// in [M,N] (ld_in) -> out_t [N, ldt] with out_t[n][m] = in[m][n]; columns m in [M, ldt) are
// written as zeros so the token dimension can serve as a padded GEMM reduction dimension.
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void transpose_kernel(const TI *__restrict__ in, long ld_in,
                                                        TO *__restrict__ out_t, long ldt,
                                                        TO *__restrict__ out_rm, long ld_rm,
                                                        float *__restrict__ colsum, int M, int N) {
    __shared__ float tile[64][65];
    const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;      // 64 x 4
    float csum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = ty * 16 + i;
        const int m = m0 + r, n = n0 + tx;
        float v = 0.f;
        if (m < M && n < N) {
            v = to_f32<TI>(in[(long)m * ld_in + n]);
            if (out_rm) out_rm[(long)m * ld_rm + n] = from_f32<TO>(v);
        }
        tile[r][tx] = v;
        csum += v;
    }
    if (colsum) {
        __shared__ float part[4][64];
        part[ty][tx] = csum;
        __syncthreads();
        if (ty == 0 && n0 + tx < N)
            atomicAdd(colsum + n0 + tx, (part[0][tx] + part[1][tx]) + (part[2][tx] + part[3][tx]));
    } else {
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = ty * 16 + i;                // column of the input tile = row of the output
        const int n = n0 + c, m = m0 + tx;
        if (n < N && m < ldt) out_t[(long)n * ldt + m] = from_f32<TO>(tile[tx][c]);
    }
}

// ---- many transposes in one launch ---------------------------------------------------------------
// The data-gradient products read every weight matrix transposed; the matrices change once per optimiser
// step, so all of them are transposed by one launch per backward pass instead of one small launch each
// (a 768 x 768 matrix is 144 tiles: such launches are latency, not bandwidth).  Entry e: src [rows, cols]
// dense -> dst [cols, rows] dense, rows and cols multiples of 64; 64 x 64 tiles, 16 bytes per lane both ways.
constexpr int kTbMax = 64;       // 1.8 KB of kernel arguments
struct TransposeBatch {
    const void *src[kTbMax];
    void *dst[kTbMax];
    int rows[kTbMax], cols[kTbMax];
    int first_tile[kTbMax + 1];
    int n;
};
template <typename T>
__global__ __launch_bounds__(256) void transpose_batch_kernel(const TransposeBatch b) {
    constexpr int V = 16 / sizeof(T);              // elements per 16-byte piece
    constexpr int CPR = 64 / V;                    // pieces per tile row
    __shared__ T tile[64][64 + 4 / sizeof(T)];     // row stride 4 bytes past 64 elements
    int e = 0;
    while (e + 1 < b.n && (int)blockIdx.x >= b.first_tile[e + 1]) ++e;
    const int local = blockIdx.x - b.first_tile[e];
    const int rows = b.rows[e], cols = b.cols[e];
    const int tiles_c = cols >> 6;
    const int r0 = (local / tiles_c) << 6, c0 = (local % tiles_c) << 6;
    const T *src = static_cast<const T *>(b.src[e]);
    T *dst = static_cast<T *>(b.dst[e]);
    typedef __attribute__((ext_vector_type(4))) unsigned u4;
#pragma unroll
    for (int i = 0; i < CPR * 64 / 256; ++i) {
        const int piece = threadIdx.x + i * 256;
        const int r = piece / CPR, c = piece % CPR;
        const u4 v = *reinterpret_cast<const u4 *>(src + (long)(r0 + r) * cols + c0 + c * V);
        T tmp[V];
        __builtin_memcpy(tmp, &v, 16);
#pragma unroll
        for (int j = 0; j < V; ++j) tile[r][c * V + j] = tmp[j];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < CPR * 64 / 256; ++i) {
        const int piece = threadIdx.x + i * 256;
        const int n = piece / CPR, m = piece % CPR;      // output row (input column), piece of V input rows
        T tmp[V];
#pragma unroll
        for (int j = 0; j < V; ++j) tmp[j] = tile[m * V + j][n];
        u4 v;
        __builtin_memcpy(&v, tmp, 16);
        *reinterpret_cast<u4 *>(dst + (long)(c0 + n) * rows + r0 + m * V) = v;
    }
}

__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float *__restrict__ slabs,
                                                           float *__restrict__ out, int S, long n,
                                                           int accumulate) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    f32x4 acc = accumulate ? *reinterpret_cast<const f32x4 *>(out + i) : f32x4{0, 0, 0, 0};
    for (int s = 0; s < S; ++s) acc += *reinterpret_cast<const f32x4 *>(slabs + (long)s * n + i);
    *reinterpret_cast<f32x4 *>(out + i) = acc;
}

// ---- LayerNorm backward ----------------------------------------------------------------------
// One row per wave and iteration, three dependent wave reductions per row: the kernel lives on
// waves in flight, so a workgroup is 8 waves (two workgroups, 16 waves per CU) while the number
// of workgroups, and with it the global atomics at the end, stays small.
constexpr int kLnbWaves = 8;
template <typename TX, typename TDY, int NCH>
__global__ __launch_bounds__(64 * kLnbWaves) void layernorm_bwd_kernel(const TX *__restrict__ x, long ldx,
                                                            const float *__restrict__ gamma,
                                                            const TDY *__restrict__ dy, long lddy,
                                                            float *__restrict__ dx, int accumulate,
                                                            float *__restrict__ dgamma,
                                                            float *__restrict__ dbeta,
                                                            uint16_t *__restrict__ dx_bf16,
                                                            float *__restrict__ dx_colsum, int M, int E,
                                                            float eps) {
    __shared__ float red[3][NCH * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 3 * NCH * 256; i += 64 * kLnbWaves) (&red[0][0])[i] = 0.f;
    __syncthreads();
    f32x4 g_[NCH], dg[NCH], db[NCH], dc[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = c * 256 + lane * 4;
        g_[c] = col < E ? *reinterpret_cast<const f32x4 *>(gamma + col) : f32x4{0, 0, 0, 0};
        dg[c] = f32x4{0, 0, 0, 0};
        db[c] = f32x4{0, 0, 0, 0};
        dc[c] = f32x4{0, 0, 0, 0};
    }
    for (int row = blockIdx.x * kLnbWaves + wave; row < M; row += gridDim.x * kLnbWaves) {
        f32x4 xv[NCH], dyv[NCH];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int col = c * 256 + lane * 4;
            if (col < E) {
                const TX *px = x + (long)row * ldx + col;
                if constexpr (sizeof(TX) == 4) {
                    xv[c] = *reinterpret_cast<const f32x4 *>(px);
                } else {
                    const u32x2 w = *reinterpret_cast<const u32x2 *>(px);
                    xv[c] = f32x4{tc_bf16_lo(w[0]), tc_bf16_hi(w[0]), tc_bf16_lo(w[1]), tc_bf16_hi(w[1])};
                }
                const TDY *pdy = dy + (long)row * lddy + col;
                if constexpr (sizeof(TDY) == 4) {
                    dyv[c] = *reinterpret_cast<const f32x4 *>(pdy);
                } else {
                    const u32x2 w = *reinterpret_cast<const u32x2 *>(pdy);
                    dyv[c] = f32x4{tc_bf16_lo(w[0]), tc_bf16_hi(w[0]), tc_bf16_lo(w[1]), tc_bf16_hi(w[1])};
                }
            } else {
                xv[c] = f32x4{0, 0, 0, 0};
                dyv[c] = f32x4{0, 0, 0, 0};
            }
            s += (xv[c][0] + xv[c][1]) + (xv[c][2] + xv[c][3]);
        }
        const float mean = tc_wave_sum(s) / (float)E;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int col = c * 256 + lane * 4;
            if (col < E) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = xv[c][j] - mean;
                    q += d * d;
                }
            }
        }
        const float rstd = rsqrtf(tc_wave_sum(q) / (float)E + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int col = c * 256 + lane * 4;
            if (col < E) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xh = (xv[c][j] - mean) * rstd;
                    const float gy = dyv[c][j] * g_[c][j];
                    xv[c][j] = xh;
                    dg[c][j] += dyv[c][j] * xh;
                    db[c][j] += dyv[c][j];
                    dyv[c][j] = gy;
                    s1 += gy;
                    s2 += gy * xh;
                }
            }
        }
        const float m1 = tc_wave_sum(s1) / (float)E, m2 = tc_wave_sum(s2) / (float)E;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int col = c * 256 + lane * 4;
            if (col < E) {
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = rstd * (dyv[c][j] - m1 - xv[c][j] * m2);
                float *po = dx + (long)row * E + col;
                if (accumulate) o += *reinterpret_cast<const f32x4 *>(po);
                *reinterpret_cast<f32x4 *>(po) = o;
                if (dx_bf16) {       // what the next Linear's gradients read: bf16 copy + column sums of dx
                    u32x2 w;
                    w[0] = tc_pack_bf16(o[0], o[1]);
                    w[1] = tc_pack_bf16(o[2], o[3]);
                    *reinterpret_cast<u32x2 *>(dx_bf16 + (long)row * E + col) = w;
                    dc[c] += o;
                }
            }
        }
    }
    // workgroup reduction of the affine gradients, then one atomic per column
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            atomicAdd(&red[0][c * 256 + lane * 4 + j], dg[c][j]);
            atomicAdd(&red[1][c * 256 + lane * 4 + j], db[c][j]);
            if (dx_colsum) atomicAdd(&red[2][c * 256 + lane * 4 + j], dc[c][j]);
        }
    __syncthreads();
    for (int col = threadIdx.x; col < E; col += 64 * kLnbWaves) {
        atomicAdd(dgamma + col, red[0][col]);
        atomicAdd(dbeta + col, red[1][col]);
        if (dx_colsum) atomicAdd(dx_colsum + col, red[2][col]);
    }
}

// ---- GELU backward (elementwise) ---------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void dgelu_kernel(const float *__restrict__ dy, const T *__restrict__ u,
                                                    T *__restrict__ out, long n) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    const f32x4 g = *reinterpret_cast<const f32x4 *>(dy + i);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float x = to_f32<T>(u[i + j]);
        const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
        const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
        out[i + j] = from_f32<T>(g[j] * fmaf(x, pdf, cdf));
    }
}

// ---- thin head backward ------------------------------------------------------------------------
// logits = LN(g) W^T + b with W [V,E]:  d_ln = dlogits W,  dW += dlogits^T LN(g),  db += colsum.
template <typename T, int NCH, int V>
__global__ __launch_bounds__(256) void head_bwd_kernel(const T *__restrict__ g, int ldg, int col0,
                                                       const float *__restrict__ gamma,
                                                       const float *__restrict__ beta,
                                                       const float *__restrict__ W,
                                                       const float *__restrict__ dlogits,
                                                       float *__restrict__ d_ln, float *__restrict__ dW,
                                                       float *__restrict__ dbias, int M, int E, float eps) {
    __shared__ float red[V][NCH * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < V * NCH * 256; i += 256) (&red[0][0])[i] = 0.f;
    __syncthreads();
    f32x4 gm[NCH], bt[NCH], acc[V][NCH];
    float accb = 0.f;                                  // lane v < V keeps dbias[v]
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = c * 256 + lane * 4;
        gm[c] = col < E ? *reinterpret_cast<const f32x4 *>(gamma + col) : f32x4{0, 0, 0, 0};
        bt[c] = col < E ? *reinterpret_cast<const f32x4 *>(beta + col) : f32x4{0, 0, 0, 0};
#pragma unroll
        for (int v = 0; v < V; ++v) acc[v][c] = f32x4{0, 0, 0, 0};
    }
    for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
        const T *src = g + (long)row * ldg + col0;
        f32x4 xv[NCH];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int col = c * 256 + lane * 4;
            xv[c] = col < E ? f32x4{to_f32<T>(src[col]), to_f32<T>(src[col + 1]), to_f32<T>(src[col + 2]),
                                    to_f32<T>(src[col + 3])}
                            : f32x4{0, 0, 0, 0};
            s += (xv[c][0] + xv[c][1]) + (xv[c][2] + xv[c][3]);
        }
        const float mean = tc_wave_sum(s) / (float)E;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int col = c * 256 + lane * 4;
            if (col < E) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = xv[c][j] - mean;
                    q += d * d;
                }
            }
        }
        const float rstd = rsqrtf(tc_wave_sum(q) / (float)E + eps);
        float dl[V];
#pragma unroll
        for (int v = 0; v < V; ++v) dl[v] = dlogits[(long)row * V + v];
        if (lane < V) accb += dlogits[(long)row * V + lane];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int col = c * 256 + lane * 4;
            if (col < E) {
                f32x4 ln, o = f32x4{0, 0, 0, 0};
#pragma unroll
                for (int j = 0; j < 4; ++j) ln[j] = (xv[c][j] - mean) * rstd * gm[c][j] + bt[c][j];
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const f32x4 wv = *reinterpret_cast<const f32x4 *>(W + (long)v * E + col);
                    o += wv * dl[v];
                    acc[v][c] += ln * dl[v];
                }
                *reinterpret_cast<f32x4 *>(d_ln + (long)row * E + col) = o;
            }
        }
    }
#pragma unroll
    for (int v = 0; v < V; ++v)
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) atomicAdd(&red[v][c * 256 + lane * 4 + j], acc[v][c][j]);
    __syncthreads();
    for (int idx = threadIdx.x; idx < V * E; idx += 256) {
        const int v = idx / E, col = idx - v * E;
        atomicAdd(dW + idx, red[v][col]);
    }
    if (lane < V) atomicAdd(dbias + lane, accb);
}

// ---- embedding backward -------------------------------------------------------------------------
// One workgroup per sequence: sums dx over the sequence's non-pad rows per token id (in LDS, each
// thread owns its columns) and over all non-pad rows for the species row.
constexpr int kEmbedBwdRows = 128;   // positions of one sequence per workgroup
__global__ __launch_bounds__(256) void embed_bwd_kernel(const int32_t *__restrict__ tokens,
                                                        const int32_t *__restrict__ species,
                                                        const float *__restrict__ dx,
                                                        float *__restrict__ d_tok,
                                                        float *__restrict__ d_sp, int T, int E, int vocab,
                                                        int n_species, int pad_idx) {
    extern __shared__ float s_tok[];                 // [vocab][E]
    const int b = blockIdx.x;
    const int t_begin = blockIdx.y * kEmbedBwdRows, t_end = min(T, t_begin + kEmbedBwdRows);
    for (int i = threadIdx.x; i < vocab * E; i += 256) s_tok[i] = 0.f;
    __syncthreads();
    const int sp = species[b];
    for (int col = threadIdx.x; col < E; col += 256) {
        float ssum = 0.f;
        for (int t = t_begin; t < t_end; ++t) {
            const int tok = tokens[(long)b * T + t];
            if (tok == pad_idx || tok < 0 || tok >= vocab) continue;
            const float v = dx[((long)b * T + t) * E + col];
            s_tok[tok * E + col] += v;
            ssum += v;
        }
        if (sp >= 0 && sp < n_species) atomicAdd(d_sp + (long)sp * E + col, ssum);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < vocab * E; i += 256) {
        const float v = s_tok[i];
        if (v != 0.f) atomicAdd(d_tok + i, v);
    }
}

// ---- rotary backward (in place on the q and k sections of d_qkv) -----------------------------------
template <typename T>
__global__ __launch_bounds__(256) void rope_bwd_kernel(T *__restrict__ dqkv, long ld,
                                                       const float *__restrict__ cs,
                                                       const float *__restrict__ sn, int M, int Tn, int E,
                                                       int head_dim, float q_scale) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;     // pair index over [M, E] (q and k sections)
    const long pairs_per_row = E;                               // 2E columns / 2
    if (idx >= (long)M * pairs_per_row) return;
    const int m = (int)(idx / pairs_per_row);
    const int pc = (int)(idx - (long)m * pairs_per_row);        // pair column in [0, E)
    const int col = pc * 2;                                     // column in [0, 2E)
    const int pi = (col % head_dim) >> 1;
    const long t = (long)(m % Tn) * (head_dim >> 1) + pi;
    const float c = cs[t], s = sn[t];
    T *p = dqkv + (long)m * ld + col;
    const float d0 = to_f32<T>(p[0]), d1 = to_f32<T>(p[1]);
    const float sc = col < E ? q_scale : 1.0f;
    p[0] = from_f32<T>((d0 * c + d1 * s) * sc);
    p[1] = from_f32<T>((d1 * c - d0 * s) * sc);
}

// ---- masked-LM objective: combine the two cross-entropy sums, scale the logit gradients -------
// stats: [0] = lm loss sum, [1] = struct loss sum (f32); counts: [0], [1] (int32)
__global__ __launch_bounds__(256) void mlm_finalize_kernel(const float *__restrict__ stats,
                                                           const int32_t *__restrict__ counts, float *loss,
                                                           float *dl, long n_dl, float *ds, long n_ds,
                                                           float struct_weight, float upstream) {
    const int c0 = counts[0], c1 = counts[1];
    const float s0 = c0 > 0 ? 1.0f / (float)c0 : 0.f;
    const float s1 = c1 > 0 ? struct_weight / (float)c1 : 0.f;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i == 0) {
        float v = c0 > 0 ? stats[0] * s0 : __builtin_nanf("");   // mean over zero rows is NaN in torch too
        if (c1 > 0) v += stats[1] * s1;
        loss[0] = v;
    }
    if (dl && i < n_dl) dl[i] *= s0 * upstream;
    if (ds && i < n_ds) ds[i] *= s1 * upstream;
}

// ---- AdamW (decoupled weight decay), torch.optim.AdamW update order ----------------------------
__global__ __launch_bounds__(256) void adamw_kernel(float *__restrict__ p, const float *__restrict__ g,
                                                    float *__restrict__ m, float *__restrict__ v, long n, float lr,
                                                    float beta1, float beta2, float eps, float wd, float bc1,
                                                    float rsqrt_bc2, float grad_scale) {
    const long i4 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 >= n) return;
    if (i4 + 4 <= n) {
        f32x4 pp = *reinterpret_cast<f32x4 *>(p + i4);
        f32x4 mm = *reinterpret_cast<f32x4 *>(m + i4);
        f32x4 vv = *reinterpret_cast<f32x4 *>(v + i4);
        const f32x4 gg = *reinterpret_cast<const f32x4 *>(g + i4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gr = gg[j] * grad_scale;
            float x = pp[j] * (1.0f - lr * wd);
            mm[j] = mm[j] + (gr - mm[j]) * (1.0f - beta1);
            vv[j] = vv[j] * beta2 + gr * gr * (1.0f - beta2);
            const float denom = sqrtf(vv[j]) * rsqrt_bc2 + eps;
            pp[j] = x - (lr / bc1) * (mm[j] / denom);
        }
        *reinterpret_cast<f32x4 *>(p + i4) = pp;
        *reinterpret_cast<f32x4 *>(m + i4) = mm;
        *reinterpret_cast<f32x4 *>(v + i4) = vv;
    } else {
        for (long i = i4; i < n; ++i) {
            const float gr = g[i] * grad_scale;
            const float x = p[i] * (1.0f - lr * wd);
            m[i] = m[i] + (gr - m[i]) * (1.0f - beta1);
            v[i] = v[i] * beta2 + gr * gr * (1.0f - beta2);
            p[i] = x - (lr / bc1) * (m[i] / (sqrtf(v[i]) * rsqrt_bc2 + eps));
        }
    }
}

// ---- masked-LM corruption of a padded batch (pipeline.py:143-195) ----------------------------------
__device__ __forceinline__ uint32_t mix32(uint32_t x) {     // finaliser of a counter-based generator
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// One workgroup per sequence.  Every eligible position (all but the first and last token) draws a
// 32-bit key; the num_changed smallest keys are the selected positions (a uniform sample without
// replacement), ordered by key: the first num_mask become <mask>, the next num_leave stay, the
// rest get a uniformly drawn coding token.
__global__ __launch_bounds__(256) void mlm_mask_kernel(const int32_t *__restrict__ tokens,
                                                       const int32_t *__restrict__ lens, int32_t *__restrict__ out,
                                                       int32_t *__restrict__ labels, int32_t *__restrict__ struct_io,
                                                       int T, uint64_t seed, double mask_prop, double mask_pct,
                                                       double leave_pct, int mask_idx, int n_coding,
                                                       const int32_t *__restrict__ coding) {
    extern __shared__ uint32_t keys[];
    const int b = blockIdx.x;
    int n = lens[b];
    n = n < 0 ? 0 : (n > T ? T : n);
    int n_changed = (int)((double)n * mask_prop);
    const int n_mask = (int)((double)n_changed * mask_pct);
    const int n_leave = (int)((double)n_changed * leave_pct);
    const int eligible = n - 2 > 0 ? n - 2 : 0;
    if (n_changed > eligible) n_changed = eligible;
    const uint32_t s0 = mix32((uint32_t)seed ^ mix32((uint32_t)(seed >> 32) + 0x9e3779b9u * (uint32_t)(b + 1)));
    for (int t = threadIdx.x; t < eligible; t += 256) keys[t] = mix32(s0 + 0x85ebca6bu * (uint32_t)(t + 1));
    __syncthreads();
    const long row = (long)b * T;
    for (int t = threadIdx.x; t < T; t += 256) {
        int tok = tokens[row + t];
        int label = -100;
        bool selected = false;
        if (t >= 1 && t - 1 < eligible) {
            const uint32_t k = keys[t - 1];
            int rank = 0;
            for (int u = 0; u < eligible; ++u) {
                const uint32_t ku = keys[u];
                rank += (ku < k) || (ku == k && u < t - 1);
            }
            if (rank < n_changed) {
                selected = true;
                label = tok;
                if (rank < n_mask) tok = mask_idx;
                else if (rank >= n_mask + n_leave) tok = coding[mix32(k ^ 0xc2b2ae35u) % (uint32_t)n_coding];
            }
        }
        out[row + t] = tok;
        labels[row + t] = label;
        if (struct_io && !selected) struct_io[row + t] = -100;
    }
}

// ---- crop + right-pad of an int batch (DataTrimmer / DataPadder, pipeline.py:250-263, 300-309) --------
__global__ __launch_bounds__(256) void crop_pad_kernel(const int32_t *__restrict__ in, int32_t *__restrict__ out,
                                                       const int32_t *__restrict__ start,
                                                       const int32_t *__restrict__ lens, int32_t *lens_out, int T_in,
                                                       int T_out, int fill) {
    const int b = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T_out) return;
    int n = lens[b];
    n = n < 0 ? 0 : (n > T_in ? T_in : n);
    int s0 = start ? start[b] : 0;
    const int keep = n < T_out ? n : T_out;
    s0 = s0 < 0 ? 0 : (s0 > n - keep ? n - keep : s0);
    out[(long)b * T_out + t] = t < keep ? in[(long)b * T_in + s0 + t] : fill;
    if (t == 0 && lens_out) lens_out[b] = keep;
}

}  // namespace

extern "C" int tc_transpose_cast(int in_dtype, int out_dtype, const void *in, int ld_in, void *out_t, int ldt,
                                 void *out_rowmajor, int ld_rm, float *colsum, int M, int N, void *stream) {
    TC_REQUIRE(in && out_t && M > 0 && N > 0 && ld_in >= N && ldt >= M, "tc_transpose_cast: bad arguments");
    TC_REQUIRE((in_dtype == TC_F32 || in_dtype == TC_BF16) && (out_dtype == TC_F32 || out_dtype == TC_BF16),
               "tc_transpose_cast: bad dtype");
    TC_REQUIRE(!(in_dtype == TC_BF16 && out_dtype == TC_F32), "tc_transpose_cast: bf16 -> f32 is not built");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((ldt + 63) / 64, (N + 63) / 64), block(256);
    if (in_dtype == TC_F32 && out_dtype == TC_F32)
        TC_LAUNCH((transpose_kernel<float, float>), grid, block, 0, s, static_cast<const float *>(in),
                           (long)ld_in, static_cast<float *>(out_t), (long)ldt, static_cast<float *>(out_rowmajor),
                           (long)ld_rm, colsum, M, N);
    else if (in_dtype == TC_F32)
        TC_LAUNCH((transpose_kernel<float, uint16_t>), grid, block, 0, s, static_cast<const float *>(in),
                           (long)ld_in, static_cast<uint16_t *>(out_t), (long)ldt,
                           static_cast<uint16_t *>(out_rowmajor), (long)ld_rm, colsum, M, N);
    else
        TC_LAUNCH((transpose_kernel<uint16_t, uint16_t>), grid, block, 0, s,
                           static_cast<const uint16_t *>(in), (long)ld_in, static_cast<uint16_t *>(out_t), (long)ldt,
                           static_cast<uint16_t *>(out_rowmajor), (long)ld_rm, colsum, M, N);
    TC_CHECK_LAUNCH("tc_transpose_cast");
    return TC_OK;
}

extern "C" int tc_transpose_batch(int dtype, int n, const void *const *src, void *const *dst, const int *rows,
                                  const int *cols, void *stream) {
    TC_REQUIRE(dtype == TC_F32 || dtype == TC_BF16, "tc_transpose_batch: bad dtype %d", dtype);
    TC_REQUIRE(n > 0 && src && dst && rows && cols, "tc_transpose_batch: bad arguments");
    for (int e = 0; e < n; ++e)
        TC_REQUIRE(src[e] && dst[e] && rows[e] > 0 && cols[e] > 0 && rows[e] % 64 == 0 && cols[e] % 64 == 0 &&
                       tc_aligned16(src[e]) && tc_aligned16(dst[e]) && (long)(rows[e] >> 6) * (cols[e] >> 6) < (1L << 24),
                   "tc_transpose_batch: entry %d: %d x %d (multiples of 64, 16-byte aligned, dense)", e, rows[e], cols[e]);
    hipStream_t s = static_cast<hipStream_t>(stream);
    for (int e0 = 0; e0 < n; e0 += kTbMax) {
        TransposeBatch b{};
        b.n = n - e0 < kTbMax ? n - e0 : kTbMax;
        int tiles = 0;
        for (int e = 0; e < b.n; ++e) {
            b.src[e] = src[e0 + e];
            b.dst[e] = dst[e0 + e];
            b.rows[e] = rows[e0 + e];
            b.cols[e] = cols[e0 + e];
            b.first_tile[e] = tiles;
            tiles += (rows[e0 + e] >> 6) * (cols[e0 + e] >> 6);
        }
        b.first_tile[b.n] = tiles;
        if (dtype == TC_BF16) TC_LAUNCH(transpose_batch_kernel<uint16_t>, dim3((unsigned)tiles), dim3(256), 0, s, b);
        else TC_LAUNCH(transpose_batch_kernel<float>, dim3((unsigned)tiles), dim3(256), 0, s, b);
        TC_CHECK_LAUNCH("tc_transpose_batch");
    }
    return TC_OK;
}

extern "C" int tc_reduce_slabs(const float *slabs, float *out, int S, long n, int accumulate, void *stream) {
    TC_REQUIRE(slabs && out && S >= 1 && n > 0 && n % 4 == 0, "tc_reduce_slabs: bad arguments");
    TC_REQUIRE(tc_aligned16(slabs) && tc_aligned16(out), "tc_reduce_slabs: misaligned operand");
    TC_LAUNCH(reduce_slabs_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), slabs, out, S, n, accumulate);
    TC_CHECK_LAUNCH("tc_reduce_slabs");
    return TC_OK;
}

extern "C" int tc_layernorm_bwd_dy(int x_dtype, const void *x, int ldx, const float *gamma, int dy_dtype,
                                   const void *dy, int lddy, float *dx, int accumulate, float *dgamma, float *dbeta,
                                   void *dx_bf16, float *dx_colsum, int M, int E, float eps, void *stream) {
    TC_REQUIRE(x && gamma && dy && dx && dgamma && dbeta, "tc_layernorm_bwd: null operand");
    TC_REQUIRE(M > 0 && E > 0 && E % 4 == 0 && E <= 1024 && ldx >= E && lddy >= E,
               "tc_layernorm_bwd: bad shape M=%d E=%d", M, E);
    TC_REQUIRE((x_dtype == TC_F32 || x_dtype == TC_BF16) && (dy_dtype == TC_F32 || dy_dtype == TC_BF16),
               "tc_layernorm_bwd: bad dtype");
    TC_REQUIRE(ldx % 4 == 0 && lddy % 4 == 0 && ((uintptr_t)x % (x_dtype == TC_F32 ? 16 : 8)) == 0 &&
                   ((uintptr_t)dy % (dy_dtype == TC_F32 ? 16 : 8)) == 0 && tc_aligned16(dx),
               "tc_layernorm_bwd: x and dy rows must be 4-element aligned, dx 16-byte aligned");
    TC_REQUIRE((dx_colsum == nullptr || dx_bf16 != nullptr) && ((uintptr_t)dx_bf16 & 7u) == 0,
               "tc_layernorm_bwd: the column sums come with the bf16 copy (8-byte aligned)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    // Waves in flight, not workgroup shape, set the pace (M = 65536, E = 768, accumulating; the same
    // totals as 4-, 8- or 16-wave workgroups give the same times): 2048 waves 169 us with f32 dy and
    // 187 us with bf16 dy, 4096 waves 182 / 160 us, 8192 waves 206 / 190 us, 1024 waves 245 / 290 us.
    // Every workgroup also ends with 2E global atomics onto the same addresses.
    const int cap = dy_dtype == TC_F32 ? 256 : 512;
    int blocks = (M + kLnbWaves - 1) / kLnbWaves;
    if (blocks > cap) blocks = cap;
    const int nch = (E + 255) / 256;
#define TC_LNB(TX, TDY, N)                                                                                      \
    TC_LAUNCH((layernorm_bwd_kernel<TX, TDY, N>), dim3(blocks), dim3(64 * kLnbWaves), 0, s, static_cast<const TX *>(x), \
              (long)ldx, gamma, static_cast<const TDY *>(dy), (long)lddy, dx, accumulate, dgamma, dbeta,           \
              static_cast<uint16_t *>(dx_bf16), dx_colsum, M, E, eps)
#define TC_LNB_N(TX, TDY)                                                                          \
    switch (nch) { case 1: TC_LNB(TX, TDY, 1); break; case 2: TC_LNB(TX, TDY, 2); break;           \
                   case 3: TC_LNB(TX, TDY, 3); break; default: TC_LNB(TX, TDY, 4); break; }
    if (x_dtype == TC_F32) {
        if (dy_dtype == TC_F32) { TC_LNB_N(float, float) } else { TC_LNB_N(float, uint16_t) }
    } else {
        if (dy_dtype == TC_F32) { TC_LNB_N(uint16_t, float) } else { TC_LNB_N(uint16_t, uint16_t) }
    }
#undef TC_LNB_N
#undef TC_LNB
    TC_CHECK_LAUNCH("tc_layernorm_bwd");
    return TC_OK;
}

extern "C" int tc_layernorm_bwd(int x_dtype, const void *x, int ldx, const float *gamma, const float *dy,
                                float *dx, int accumulate, float *dgamma, float *dbeta, int M, int E, float eps,
                                void *stream) {
    return tc_layernorm_bwd_dy(x_dtype, x, ldx, gamma, TC_F32, dy, E, dx, accumulate, dgamma, dbeta, nullptr, nullptr, M,
                               E, eps, stream);
}

extern "C" int tc_dgelu(int dtype, const float *dy, const void *u, void *out, long n, void *stream) {
    TC_REQUIRE(dy && u && out && n > 0 && n % 4 == 0, "tc_dgelu: bad arguments");
    TC_REQUIRE(dtype == TC_F32 || dtype == TC_BF16, "tc_dgelu: bad dtype");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)((n / 4 + 255) / 256)), block(256);
    if (dtype == TC_F32)
        TC_LAUNCH(dgelu_kernel<float>, grid, block, 0, s, dy, static_cast<const float *>(u),
                           static_cast<float *>(out), n);
    else
        TC_LAUNCH(dgelu_kernel<uint16_t>, grid, block, 0, s, dy, static_cast<const uint16_t *>(u),
                           static_cast<uint16_t *>(out), n);
    TC_CHECK_LAUNCH("tc_dgelu");
    return TC_OK;
}

extern "C" int tc_head_logits_bwd(int dtype, const void *g, int ldg, int col0, const float *gamma,
                                  const float *beta, const float *W, const float *dlogits, float *d_ln,
                                  float *dW, float *dbias, int M, int E, int V, float eps, void *stream) {
    TC_REQUIRE(g && gamma && beta && W && dlogits && d_ln && dW && dbias, "tc_head_logits_bwd: null operand");
    TC_REQUIRE(M > 0 && E > 0 && E % 4 == 0 && E <= 768 && (V == 3 || V == 10),
               "tc_head_logits_bwd: unsupported shape E=%d V=%d (E <= 768, V in {3,10})", E, V);
    TC_REQUIRE(dtype == TC_F32 || dtype == TC_BF16, "tc_head_logits_bwd: bad dtype");
    hipStream_t s = static_cast<hipStream_t>(stream);
    int blocks = (M + 3) / 4;
    if (blocks > 512) blocks = 512;
    const int nch = (E + 255) / 256;
#define TC_HB(TY, N, VV)                                                                                   \
    TC_LAUNCH((head_bwd_kernel<TY, N, VV>), dim3(blocks), dim3(256), 0, s, static_cast<const TY *>(g), \
                       ldg, col0, gamma, beta, W, dlogits, d_ln, dW, dbias, M, E, eps)
#define TC_HB_V(TY, N) do { if (V == 3) TC_HB(TY, N, 3); else TC_HB(TY, N, 10); } while (0)
    if (dtype == TC_F32) {
        switch (nch) { case 1: TC_HB_V(float, 1); break; case 2: TC_HB_V(float, 2); break; default: TC_HB_V(float, 3); break; }
    } else {
        switch (nch) { case 1: TC_HB_V(uint16_t, 1); break; case 2: TC_HB_V(uint16_t, 2); break; default: TC_HB_V(uint16_t, 3); break; }
    }
#undef TC_HB_V
#undef TC_HB
    TC_CHECK_LAUNCH("tc_head_logits_bwd");
    return TC_OK;
}

extern "C" int tc_embed_bwd(const int32_t *tokens, const int32_t *species, const float *dx, float *d_tok_table,
                            float *d_species_table, int B, int T, int E, int vocab, int n_species, int pad_idx,
                            void *stream) {
    TC_REQUIRE(tokens && species && dx && d_tok_table && d_species_table, "tc_embed_bwd: null operand");
    TC_REQUIRE(B > 0 && T > 0 && E > 0 && vocab > 0 && (size_t)vocab * E * 4 <= 64 * 1024,
               "tc_embed_bwd: bad shape (vocab*E*4 must fit 64 KiB of LDS)");
    TC_LAUNCH(embed_bwd_kernel, dim3(B, (T + kEmbedBwdRows - 1) / kEmbedBwdRows), dim3(256), (size_t)vocab * E * 4, static_cast<hipStream_t>(stream),
                       tokens, species, dx, d_tok_table, d_species_table, T, E, vocab, n_species, pad_idx);
    TC_CHECK_LAUNCH("tc_embed_bwd");
    return TC_OK;
}

extern "C" int tc_rope_bwd(int dtype, void *dqkv, int ld, const float *rope_cos, const float *rope_sin, int M,
                           int T, int E, int head_dim, float q_scale, void *stream) {
    TC_REQUIRE(dqkv && rope_cos && rope_sin && M > 0 && T > 0 && E > 0 && head_dim > 0 && head_dim % 2 == 0 &&
                   E % head_dim == 0 && ld >= 2 * E,
               "tc_rope_bwd: bad arguments");
    TC_REQUIRE(dtype == TC_F32 || dtype == TC_BF16, "tc_rope_bwd: bad dtype");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long n = (long)M * E;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (dtype == TC_F32)
        TC_LAUNCH(rope_bwd_kernel<float>, grid, block, 0, s, static_cast<float *>(dqkv), (long)ld, rope_cos,
                           rope_sin, M, T, E, head_dim, q_scale);
    else
        TC_LAUNCH(rope_bwd_kernel<uint16_t>, grid, block, 0, s, static_cast<uint16_t *>(dqkv), (long)ld,
                           rope_cos, rope_sin, M, T, E, head_dim, q_scale);
    TC_CHECK_LAUNCH("tc_rope_bwd");
    return TC_OK;
}

extern "C" int tc_mlm_loss(const float *logits, const int32_t *labels, const float *struct_logits,
                           const int32_t *struct_labels, float *loss, float *dlogits, float *dstruct_logits,
                           float *stats, float struct_weight, float upstream, int M, int V, int S, void *stream) {
    TC_REQUIRE(logits && labels && loss && stats, "tc_mlm_loss: null operand");
    TC_REQUIRE((struct_logits != nullptr) == (struct_labels != nullptr), "tc_mlm_loss: struct logits and labels go together");
    TC_REQUIRE(!dstruct_logits || struct_logits, "tc_mlm_loss: struct gradient without struct logits");
    TC_REQUIRE(M > 0 && V > 0 && V <= 64 && (!struct_logits || (S > 0 && S <= 64)), "tc_mlm_loss: bad shape");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const hipError_t e = hipMemsetAsync(stats, 0, 16, s);
    if (e != hipSuccess) {
        tc_set_error("tc_mlm_loss: memset: %s", hipGetErrorString(e));
        return TC_ELAUNCH;
    }
    int32_t *counts = reinterpret_cast<int32_t *>(stats + 2);
    int rc = tc_ce_loss_fwd_bwd(logits, labels, stats, counts, dlogits, 1.0f, M, V, stream);
    if (rc != TC_OK) return rc;
    if (struct_logits) {
        rc = tc_ce_loss_fwd_bwd(struct_logits, struct_labels, stats + 1, counts + 1, dstruct_logits, 1.0f, M, S, stream);
        if (rc != TC_OK) return rc;
    }
    const long n_dl = dlogits ? (long)M * V : 0, n_ds = dstruct_logits ? (long)M * S : 0;
    long n = n_dl > n_ds ? n_dl : n_ds;
    if (n < 1) n = 1;
    TC_LAUNCH(mlm_finalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, stats, counts, loss,
                       dlogits, n_dl, dstruct_logits, n_ds, struct_weight, upstream);
    TC_CHECK_LAUNCH("tc_mlm_loss");
    return TC_OK;
}

extern "C" int tc_crop_pad(const int32_t *in, int32_t *out, const int32_t *start, const int32_t *lens,
                           int32_t *lens_out, int B, int T_in, int T_out, int fill, void *stream) {
    TC_REQUIRE(in && out && lens && in != out, "tc_crop_pad: null or aliased operand");
    TC_REQUIRE(B > 0 && B <= 65535 && T_in > 0 && T_out > 0, "tc_crop_pad: bad shape B=%d T_in=%d T_out=%d", B, T_in, T_out);
    TC_LAUNCH(crop_pad_kernel, dim3((T_out + 255) / 256, B), dim3(256), 0, static_cast<hipStream_t>(stream), in, out,
              start, lens, lens_out, T_in, T_out, fill);
    TC_CHECK_LAUNCH("tc_crop_pad");
    return TC_OK;
}

extern "C" int tc_adamw_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, long n, float lr,
                             float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                             void *stream) {
    TC_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, "tc_adamw_step: bad arguments");
    TC_REQUIRE(tc_aligned16(param) && tc_aligned16(grad) && tc_aligned16(exp_avg) && tc_aligned16(exp_avg_sq),
               "tc_adamw_step: operands must be 16-byte aligned");
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    TC_LAUNCH(adamw_kernel, dim3((unsigned)(((n + 3) / 4 + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps,
                       weight_decay, (float)bc1, (float)(1.0 / sqrt(bc2)), grad_scale);
    TC_CHECK_LAUNCH("tc_adamw_step");
    return TC_OK;
}

extern "C" int tc_mlm_mask(const int32_t *tokens, const int32_t *lens, int32_t *tokens_out, int32_t *labels,
                           int32_t *struct_labels, int B, int T, uint64_t seed, double mask_proportion,
                           double mask_percent, double leave_percent, int mask_idx, const int32_t *coding_ids,
                           int n_coding, void *stream) {
    TC_REQUIRE(tokens && lens && tokens_out && labels && coding_ids, "tc_mlm_mask: null operand");
    TC_REQUIRE(B > 0 && T > 0 && T <= 16384 && n_coding > 0, "tc_mlm_mask: bad shape B=%d T=%d (T <= 16384)", B, T);
    TC_REQUIRE(mask_proportion >= 0 && mask_proportion <= 1 && mask_percent >= 0 && leave_percent >= 0 &&
                   mask_percent + leave_percent <= 1.0,
               "tc_mlm_mask: bad proportions");
    TC_LAUNCH(mlm_mask_kernel, dim3(B), dim3(256), (size_t)T * 4, static_cast<hipStream_t>(stream), tokens, lens,
                       tokens_out, labels, struct_labels, T, seed, mask_proportion, mask_percent, leave_percent, mask_idx,
                       n_coding, coding_ids);
    TC_CHECK_LAUNCH("tc_mlm_mask");
    return TC_OK;
}
