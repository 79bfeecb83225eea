// Backward of the fused multi-head self-attention (head_dim 64, gfx950).
//
// With P = softmax(S), S = Q K^T (+ -inf at padded keys), O = P V and the incoming dO:
//   delta[q] = sum_d dO[q,d] O[q,d]            (= sum_k P[q,k] dP[q,k])
//   dP = dO V^T,  dS = P o (dP - delta),  dQ = dS K,  dK = dS^T Q,  dV = P^T dO.
// P is rebuilt from the forward's per-query log-sum-exp: P = exp(S - lse).
//
// bf16: two MFMA kernels, neither of which needs atomics.
//   dQ kernel  -- same shape as the forward: a workgroup owns 128 queries (32 per wave, query
//     on the lane), streams 64-key K/V tiles through LDS and accumulates dQ^T = K^T dS^T.
//   dK/dV kernel -- a workgroup owns 128 keys (32 per wave, key on the lane), streams 64-query
//     tiles of Q, dO, lse and delta through LDS and accumulates dK^T = Q^T dS, dV^T = dO^T P.
// Operands that are read both by rows (A operand of the S / dP products) and transposed
// (A operand of the gradient products, via ds_read_b64_tr_b16) are staged twice with the
// swizzle that suits each read pattern.
//
// f32 (exactness path): VALU kernels, one query (dQ) or one key (dK/dV) per lane.
#include "common.h"

namespace {

constexpr float kLog2e = 1.4426950408889634f;
constexpr int TILE_BYTES = 64 * 128;   // 64 rows x 64 bf16
constexpr int OST_LD = 144;
constexpr int OST_WAVE_BYTES = 32 * OST_LD;

struct BwdParams {
    const char *qkv;
    const uint64_t *words;
    const char *out;      // forward output O [B*T, E]
    const char *dout;     // dO [B*T, E]
    const float *lse;     // [B,H,T]
    float *delta;         // [B,H,T]
    char *dqkv;           // [B*T, ldg]: dQ | dK | dV
    const float *rope_cos, *rope_sin;   // optional [T, 32]: also undo the rotation (and q scale) of dQ, dK
    float q_scale;
    long ldq_bytes, ldo_bytes, ldg_bytes;
    int B, T, H, E;
    int nw;               // validity words per sequence
    int nblk;             // 128-row blocks per (b,h)
};

__device__ __forceinline__ int swz_row(int row) { return (row >> 1) & 7; }         // row reads
__device__ __forceinline__ int swz_tr(int row) { return ((row >> 1) & 1) << 2; }   // transposed reads

constexpr int kMaxWords = 128;   // key tiles whose validity words the bf16 kernels keep in LDS (T <= 8192)

// Two transposing reads, rows r and r + 8 of a 128-byte-pitch image (lo / hi half of a fragment),
// issued from assembly: through the builtin the compiler cannot tell them from the LDS-DMA writes
// of the next tile in flight and waits for that prefetch (vmcnt(0)) in the middle of the loop.
// The caller waits (tr_wait) before an MFMA reads the result.
template <int OFF>
__device__ __forceinline__ void tr_read_pair(unsigned at, bf16x4 &lo, bf16x4 &hi) {
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"
                 : "=&v"(lo), "=&v"(hi)
                 : "v"(at), "n"(OFF), "n"(OFF + 1024));
}
__device__ __forceinline__ void tr_wait(bf16x4 &a, bf16x4 &b, bf16x4 &c, bf16x4 &d) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
__device__ __forceinline__ bf16x8 tr_join(bf16x4 l, bf16x4 h) { return __builtin_shufflevector(l, h, 0, 1, 2, 3, 4, 5, 6, 7); }

__device__ __forceinline__ uint64_t key_word(const BwdParams &p, int b, int kt) {
    const int rem = p.T - kt * 64;
    const uint64_t in_range = rem >= 64 ? ~0ull : ((1ull << rem) - 1ull);
    return p.words ? (p.words[(long)b * p.nw + kt] & in_range) : in_range;
}

// Backward of the rotary rotation (and of the q scale) on 8 consecutive columns = 4 pairs of one
// head, in the bf16 staging form: (d0, d1) <- (d0 c + d1 s, d1 c - d0 s) * scale.
__device__ __forceinline__ u32x4 rope_bwd8(u32x4 w, const BwdParams &p, int pos, int chunk, float scale) {
    const f32x4 cs = *reinterpret_cast<const f32x4 *>(p.rope_cos + (long)pos * 32 + chunk * 4);
    const f32x4 sn = *reinterpret_cast<const f32x4 *>(p.rope_sin + (long)pos * 32 + chunk * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float d0 = tc_bf16_lo(w[j]), d1 = tc_bf16_hi(w[j]);
        w[j] = tc_pack_bf16((d0 * cs[j] + d1 * sn[j]) * scale, (d1 * cs[j] - d0 * sn[j]) * scale);
    }
    return w;
}

// ------------------------------------------------------------------------------------------
// delta[b,h,t] = sum_d dO * O ; 8 lanes per (row, head), 8 elements per lane
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void delta_kernel(const T *__restrict__ o, const T *__restrict__ d_o,
                                                    float *__restrict__ delta, long rows, int Tn, int H) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const long grp = gid >> 3;                 // (row, head)
    const int part = (int)(gid & 7);
    const long total = rows * H;
    const long g = grp < total ? grp : total - 1;
    const long row = g / H;
    const int h = (int)(g % H);
    const long off = row * (long)H * 64 + h * 64 + part * 8;
    float acc = 0.f;
    if constexpr (sizeof(T) == 2) {
        const u32x4 a = *reinterpret_cast<const u32x4 *>(o + off);
        const u32x4 c = *reinterpret_cast<const u32x4 *>(d_o + off);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            acc += tc_bf16_lo(a[i]) * tc_bf16_lo(c[i]) + tc_bf16_hi(a[i]) * tc_bf16_hi(c[i]);
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(o + off + 4 * i);
            const f32x4 c = *reinterpret_cast<const f32x4 *>(d_o + off + 4 * i);
            acc += a[0] * c[0] + a[1] * c[1] + a[2] * c[2] + a[3] * c[3];
        }
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    if (part == 0 && grp < total) {
        const long b = row / Tn, t = row % Tn;
        delta[(b * H + h) * Tn + t] = acc;
    }
}

// ------------------------------------------------------------------------------------------
// bf16 dQ
// ------------------------------------------------------------------------------------------
constexpr int DQ_STAGE_BYTES = 3 * TILE_BYTES;                       // K rows, K for tr reads, V rows
constexpr int DQ_LDS_BYTES = 2 * DQ_STAGE_BYTES;      // the output staging reuses the stages after the loop

__global__ __launch_bounds__(256) void attn_bwd_dq_bf16_kernel(const BwdParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int id = tc_xcd_remap(blockIdx.x, gridDim.x);
    const int qb = id % p.nblk;
    const int bh = id / p.nblk;
    const int h = bh % p.H, b = bh / p.H;
    const int ql = lane & 31, h2 = lane >> 5;
    const int q0 = qb * 128 + wid * 32;
    const long row_base = (long)b * p.T;

    // the key-validity words of this batch row go to LDS once (a global load consumed inside the
    // key loop would make the compiler wait for the K/V prefetch in the middle of an iteration)
    __shared__ uint64_t s_words[kMaxWords];
    for (int i = threadIdx.x; i < p.nw && i < kMaxWords; i += 256) s_words[i] = key_word(p, b, i);
    __syncthreads();
    int nkt = p.nw;
    while (nkt > 0 && s_words[nkt - 1] == 0ull) --nkt;

    // Q and dO fragments (B operands) and the query's lse / delta
    bf16x8 qf[4], dof[4];
    float lse2, dlt;
    {
        int qr = q0 + ql;
        const bool ok = qr < p.T;
        qr = ok ? qr : p.T - 1;
        const char *qs = p.qkv + (row_base + qr) * p.ldq_bytes + (long)h * 128 + h2 * 16;
        const char *ds = p.dout + (row_base + qr) * p.ldo_bytes + (long)h * 128 + h2 * 16;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qf[ks] = *reinterpret_cast<const bf16x8 *>(qs + ks * 32);
            dof[ks] = *reinterpret_cast<const bf16x8 *>(ds + ks * 32);
        }
        const long si = ((long)b * p.H + h) * p.T + qr;
        lse2 = ok ? p.lse[si] * kLog2e : INFINITY;
        dlt = p.delta[si];
    }

    const int srow = lane >> 3;
    // Buffer-descriptor loads over this sequence's rows: lane offset fixed, key tile scalar; only a
    // partial last tile builds per-lane offsets (rows past the sequence re-read row T-1).
    unsigned src_kr[2], src_kt[2], src_vr[2];
    int s_row[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (wid * 2 + i) * 8 + srow;
        s_row[i] = r;
        src_kr[i] = (unsigned)(p.E + h * 64) * 2u + (unsigned)(((lane & 7) ^ swz_row(r)) << 4);
        src_kt[i] = (unsigned)(p.E + h * 64) * 2u + (unsigned)(((lane & 7) ^ swz_tr(r)) << 4);
        src_vr[i] = (unsigned)(2 * p.E + h * 64) * 2u + (unsigned)(((lane & 7) ^ swz_row(r)) << 4);
    }
    const __amdgpu_buffer_rsrc_t rsrc_kv = tc_buffer_rsrc(p.qkv + row_base * p.ldq_bytes);
    const unsigned kv_row0[2] = {(unsigned)(s_row[0] * p.ldq_bytes), (unsigned)(s_row[1] * p.ldq_bytes)};
    auto stage = [&](int buf, int kt) {
        char *base = smem + buf * DQ_STAGE_BYTES + wid * 2 * 1024;
        auto pieces = [&](int i, unsigned row_off, unsigned tile_off) {
            tc_blds16(rsrc_kv, row_off + src_kr[i], tile_off, base + i * 1024);
            tc_blds16(rsrc_kv, row_off + src_kt[i], tile_off, base + TILE_BYTES + i * 1024);
            tc_blds16(rsrc_kv, row_off + src_vr[i], tile_off, base + 2 * TILE_BYTES + i * 1024);
        };
        if (kt * 64 + 64 <= p.T) {
            const unsigned tile_off = (unsigned)((long)kt * 64 * p.ldq_bytes);
#pragma unroll
            for (int i = 0; i < 2; ++i) pieces(i, kv_row0[i], tile_off);
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int key = kt * 64 + s_row[i];
                key = key < p.T ? key : p.T - 1;
                pieces(i, (unsigned)(key * p.ldq_bytes), 0u);
            }
        }
    };

    int r_off[2][4];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const int r = kb * 32 + ql;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) r_off[kb][ks] = r * 128 + (((2 * ks + h2) ^ swz_row(r)) << 4);
    }
    const int tq = (lane & 15) >> 2, tp = lane & 3, thalf = (lane >> 4) & 1;
    auto t_off = [&](int db, int row) {
        const int c = 4 * db + 2 * thalf + (tp >> 1);
        return row * 128 + ((c ^ swz_tr(row)) << 4) + 8 * (tp & 1);
    };

    // LDS byte address of this lane's transposed K reads for d block db, key rows 4 h2 + tq of a stage
    const unsigned lds_base = (unsigned)(unsigned long)(__attribute__((address_space(3))) char *)smem;
    const unsigned tr_lane[2] = {lds_base + (unsigned)t_off(0, 4 * h2 + tq), lds_base + (unsigned)t_off(1, 4 * h2 + tq)};

    f32x16 acc_dq[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_dq[db][r] = 0.f;

    if (nkt > 0) stage(0, 0);
    for (int kt = 0; kt < nkt; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nkt) stage((kt + 1) & 1, kt + 1);
        const char *sb = smem + (kt & 1) * DQ_STAGE_BYTES;
        const unsigned kt_at[2] = {tr_lane[0] + (unsigned)((kt & 1) * DQ_STAGE_BYTES),
                                   tr_lane[1] + (unsigned)((kt & 1) * DQ_STAGE_BYTES)};

        f32x16 acc_s[2], acc_p[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc_s[kb][r] = 0.f; acc_p[kb][r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8 *>(sb + r_off[kb][ks]);
                acc_s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], acc_s[kb], 0, 0, 0);
                const bf16x8 vf = *reinterpret_cast<const bf16x8 *>(sb + 2 * TILE_BYTES + r_off[kb][ks]);
                acc_p[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[ks], acc_p[kb], 0, 0, 0);
            }
        }

        // P = exp2(S log2e - lse log2e), dS = P (dP - delta): packed f32 arithmetic; padded keys get
        // S = -inf first (only in tiles that contain one: wave-uniform branch)
        const uint64_t word = s_words[kt];
        if (word != ~0ull) {
            const uint64_t wsh = word >> (4 * h2);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const uint32_t wk = (uint32_t)(wsh >> (32 * kb));
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int bit = (r & 3) + 8 * (r >> 2);
                    acc_s[kb][r] = ((wk >> bit) & 1u) ? acc_s[kb][r] : -INFINITY;
                }
            }
        }
        {
            const f32x2 scale2 = {kLog2e, kLog2e}, neg_lse2 = {-lse2, -lse2}, neg_dlt2 = {-dlt, -dlt};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2 t = f32x2{acc_s[kb][r], acc_s[kb][r + 1]} * scale2 + neg_lse2;
                    const f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
                    const f32x2 ds = e * (f32x2{acc_p[kb][r], acc_p[kb][r + 1]} + neg_dlt2);
                    acc_s[kb][r] = ds[0];
                    acc_s[kb][r + 1] = ds[1];
                }
        }

        // dQ^T[d][query] += K^T . dS^T
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8 df;
#pragma unroll
                for (int j = 0; j < 8; ++j) df[j] = (__bf16)acc_s[kb][8 * s + j];
                // rows + (32 kb + 16 s): immediates; the lane part (4 h2 + tq, swizzle) is in kt_at
                bf16x4 lo[2], hi[2];
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    switch (kb * 2 + s) {
                        case 0: tr_read_pair<TILE_BYTES>(kt_at[db], lo[db], hi[db]); break;
                        case 1: tr_read_pair<TILE_BYTES + 2048>(kt_at[db], lo[db], hi[db]); break;
                        case 2: tr_read_pair<TILE_BYTES + 4096>(kt_at[db], lo[db], hi[db]); break;
                        default: tr_read_pair<TILE_BYTES + 6144>(kt_at[db], lo[db], hi[db]); break;
                    }
                }
                tr_wait(lo[0], hi[0], lo[1], hi[1]);
#pragma unroll
                for (int db = 0; db < 2; ++db)
                    acc_dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_join(lo[db], hi[db]), df, acc_dq[db], 0, 0, 0);
            }
    }

    __syncthreads();
    char *ost = smem + wid * OST_WAVE_BYTES;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            u32x2 w;
            w[0] = tc_pack_bf16(acc_dq[db][4 * g], acc_dq[db][4 * g + 1]);
            w[1] = tc_pack_bf16(acc_dq[db][4 * g + 2], acc_dq[db][4 * g + 3]);
            const int d = 32 * db + 8 * g + 4 * h2;
            *reinterpret_cast<u32x2 *>(ost + ql * OST_LD + d * 2) = w;
        }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int r = it * 8 + (lane >> 3), c = lane & 7;
        u32x4 w = *reinterpret_cast<const u32x4 *>(ost + r * OST_LD + c * 16);
        const int q = q0 + r;
        if (q < p.T) {
            if (p.rope_cos) w = rope_bwd8(w, p, q, c, p.q_scale);
            *reinterpret_cast<u32x4 *>(p.dqkv + (row_base + q) * p.ldg_bytes + (long)h * 128 + c * 16) = w;
        }
    }
}

// ------------------------------------------------------------------------------------------
// bf16 dK / dV
// ------------------------------------------------------------------------------------------
constexpr int DKV_STAGE_BYTES = 4 * TILE_BYTES + 512;   // Q rows, Q tr, dO rows, dO tr, lse[64], delta[64]
constexpr int DKV_LDS_BYTES = 2 * DKV_STAGE_BYTES;      // output staging reuses the stage area

__global__ __launch_bounds__(256) void attn_bwd_dkv_bf16_kernel(const BwdParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int id = tc_xcd_remap(blockIdx.x, gridDim.x);
    const int kblk = id % p.nblk;
    const int bh = id / p.nblk;
    const int h = bh % p.H, b = bh / p.H;
    const int kl = lane & 31, h2 = lane >> 5;
    const int k0 = kblk * 128 + wid * 32;
    const long row_base = (long)b * p.T;
    const long stat_base = ((long)b * p.H + h) * p.T;
    const int nqt = (p.T + 63) / 64;

    // K and V fragments (B operands, key on the lane) and the key's validity
    bf16x8 kf[4], vf[4];
    bool k_ok;
    {
        int kr = k0 + kl;
        k_ok = kr < p.T;
        kr = k_ok ? kr : p.T - 1;
        k_ok = k_ok && ((key_word(p, b, kr >> 6) >> (kr & 63)) & 1ull);
        const char *ks_ = p.qkv + (row_base + kr) * p.ldq_bytes + (long)(p.E + h * 64) * 2 + h2 * 16;
        const char *vs_ = ks_ + (long)p.E * 2;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            kf[ks] = *reinterpret_cast<const bf16x8 *>(ks_ + ks * 32);
            vf[ks] = *reinterpret_cast<const bf16x8 *>(vs_ + ks * 32);
        }
    }

    const int srow = lane >> 3;
    int s_row[2];
    unsigned c_row[2], c_tr[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (wid * 2 + i) * 8 + srow;
        s_row[i] = r;
        c_row[i] = (unsigned)(h * 128 + (((lane & 7) ^ swz_row(r)) << 4));
        c_tr[i] = (unsigned)(h * 128 + (((lane & 7) ^ swz_tr(r)) << 4));
    }
    // buffer-descriptor loads as in the dQ kernel: lane offset fixed, query tile scalar
    const __amdgpu_buffer_rsrc_t rsrc_q = tc_buffer_rsrc(p.qkv + row_base * p.ldq_bytes);
    const __amdgpu_buffer_rsrc_t rsrc_d = tc_buffer_rsrc(p.dout + row_base * p.ldo_bytes);
    const unsigned q_row0[2] = {(unsigned)(s_row[0] * p.ldq_bytes), (unsigned)(s_row[1] * p.ldq_bytes)};
    const unsigned d_row0[2] = {(unsigned)(s_row[0] * p.ldo_bytes), (unsigned)(s_row[1] * p.ldo_bytes)};
    auto stage = [&](int buf, int qt) {
        char *stg = smem + buf * DKV_STAGE_BYTES;
        char *base = stg + wid * 2 * 1024;
        auto pieces = [&](int i, unsigned q_row, unsigned d_row, unsigned q_tile, unsigned d_tile) {
            tc_blds16(rsrc_q, q_row + c_row[i], q_tile, base + i * 1024);
            tc_blds16(rsrc_q, q_row + c_tr[i], q_tile, base + TILE_BYTES + i * 1024);
            tc_blds16(rsrc_d, d_row + c_row[i], d_tile, base + 2 * TILE_BYTES + i * 1024);
            tc_blds16(rsrc_d, d_row + c_tr[i], d_tile, base + 3 * TILE_BYTES + i * 1024);
        };
        if (qt * 64 + 64 <= p.T) {
            const unsigned q_tile = (unsigned)((long)qt * 64 * p.ldq_bytes), d_tile = (unsigned)((long)qt * 64 * p.ldo_bytes);
#pragma unroll
            for (int i = 0; i < 2; ++i) pieces(i, q_row0[i], d_row0[i], q_tile, d_tile);
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int q = qt * 64 + s_row[i];
                q = q < p.T ? q : p.T - 1;
                pieces(i, (unsigned)(q * p.ldq_bytes), (unsigned)(q * p.ldo_bytes), 0u, 0u);
            }
        }
        if (wid < 2) {      // wave 0: lse, wave 1: delta (one float per lane)
            int q = qt * 64 + lane;
            q = q < p.T ? q : p.T - 1;
            const float *src = (wid == 0 ? p.lse : p.delta) + stat_base + q;
            __builtin_amdgcn_global_load_lds(TC_GLOBAL_PTR(src), TC_LDS_PTR(stg + 4 * TILE_BYTES + wid * 256), 4, 0,
                                             0);
        }
    };

    int r_off[2][4];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int r = qb * 32 + kl;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) r_off[qb][ks] = r * 128 + (((2 * ks + h2) ^ swz_row(r)) << 4);
    }
    const int tq = (lane & 15) >> 2, tp = lane & 3, thalf = (lane >> 4) & 1;
    auto t_off = [&](int db, int row) {
        const int c = 4 * db + 2 * thalf + (tp >> 1);
        return row * 128 + ((c ^ swz_tr(row)) << 4) + 8 * (tp & 1);
    };

    // LDS byte address of this lane's transposed reads for d block db, query rows 4 h2 + tq of an image
    const unsigned lds_base = (unsigned)(unsigned long)(__attribute__((address_space(3))) char *)smem;
    const unsigned tr_lane[2] = {lds_base + (unsigned)t_off(0, 4 * h2 + tq), lds_base + (unsigned)t_off(1, 4 * h2 + tq)};

    f32x16 acc_dk[2], acc_dv[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc_dk[db][r] = 0.f; acc_dv[db][r] = 0.f; }

    stage(0, 0);
    for (int qt = 0; qt < nqt; ++qt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (qt + 1 < nqt) stage((qt + 1) & 1, qt + 1);
        const char *sb = smem + (qt & 1) * DKV_STAGE_BYTES;
        const unsigned tr_at[2] = {tr_lane[0] + (unsigned)((qt & 1) * DKV_STAGE_BYTES),
                                   tr_lane[1] + (unsigned)((qt & 1) * DKV_STAGE_BYTES)};
        const float *lse_s = reinterpret_cast<const float *>(sb + 4 * TILE_BYTES);
        const float *dlt_s = lse_s + 64;
        const bool ragged = qt * 64 + 64 > p.T;

        const bool select = ragged || __builtin_amdgcn_ballot_w64(!k_ok) != 0ull;   // wave-uniform
        const f32x2 scale2 = {kLog2e, kLog2e};

        // one block of 32 queries at a time keeps only 32 score registers live
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            // S[query][key] = Q K^T, dP[query][key] = dO V^T
            bf16x8 qa[4], da[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                qa[ks] = *reinterpret_cast<const bf16x8 *>(sb + r_off[qb][ks]);
                da[ks] = *reinterpret_cast<const bf16x8 *>(sb + 2 * TILE_BYTES + r_off[qb][ks]);
            }
            f32x16 acc_s, acc_p;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc_s[r] = 0.f; acc_p[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                acc_s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[ks], kf[ks], acc_s, 0, 0, 0);
                acc_p = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da[ks], vf[ks], acc_p, 0, 0, 0);
            }

            // P = exp2((S - lse) log2e) into acc_s, dS = P (dP - delta) into acc_p
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int qi = 32 * qb + 8 * g + 4 * h2;
                const f32x4 l4 = *reinterpret_cast<const f32x4 *>(lse_s + qi);
                const f32x4 d4 = *reinterpret_cast<const f32x4 *>(dlt_s + qi);
#pragma unroll
                for (int j = 0; j < 4; j += 2) {
                    const int r = 4 * g + j;
                    const f32x2 t = (f32x2{acc_s[r], acc_s[r + 1]} - f32x2{l4[j], l4[j + 1]}) * scale2;
                    const f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
                    const f32x2 ds = e * (f32x2{acc_p[r], acc_p[r + 1]} - f32x2{d4[j], d4[j + 1]});
                    acc_s[r] = e[0];
                    acc_s[r + 1] = e[1];
                    acc_p[r] = ds[0];
                    acc_p[r + 1] = ds[1];
                }
            }
            if (select) {        // padded keys of this lane / queries past the end contribute nothing
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int q = qt * 64 + 32 * qb + (r & 3) + 8 * (r >> 2) + 4 * h2;
                    const bool ok = k_ok && q < p.T;
                    acc_s[r] = ok ? acc_s[r] : 0.f;
                    acc_p[r] = ok ? acc_p[r] : 0.f;
                }
            }

            // dV^T[d][key] += dO^T . P ;  dK^T[d][key] += Q^T . dS
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8 pf, df;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    pf[j] = (__bf16)acc_s[8 * s + j];
                    df[j] = (__bf16)acc_p[8 * s + j];
                }
                // query rows + (32 qb + 16 s): immediates; the lane part (4 h2 + tq, swizzle) is in tr_at
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    bf16x4 dlo, dhi, qlo, qhi;
                    switch (qb * 2 + s) {
                        case 0:
                            tr_read_pair<3 * TILE_BYTES>(tr_at[db], dlo, dhi);
                            tr_read_pair<TILE_BYTES>(tr_at[db], qlo, qhi);
                            break;
                        case 1:
                            tr_read_pair<3 * TILE_BYTES + 2048>(tr_at[db], dlo, dhi);
                            tr_read_pair<TILE_BYTES + 2048>(tr_at[db], qlo, qhi);
                            break;
                        case 2:
                            tr_read_pair<3 * TILE_BYTES + 4096>(tr_at[db], dlo, dhi);
                            tr_read_pair<TILE_BYTES + 4096>(tr_at[db], qlo, qhi);
                            break;
                        default:
                            tr_read_pair<3 * TILE_BYTES + 6144>(tr_at[db], dlo, dhi);
                            tr_read_pair<TILE_BYTES + 6144>(tr_at[db], qlo, qhi);
                            break;
                    }
                    tr_wait(dlo, dhi, qlo, qhi);
                    acc_dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_join(dlo, dhi), pf, acc_dv[db], 0, 0, 0);
                    acc_dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_join(qlo, qhi), df, acc_dk[db], 0, 0, 0);
                }
            }
        }
    }

    // ---- epilogue: dK then dV through the wave's staging rows (stage area is free now) ----
    __syncthreads();
    char *ost = smem + wid * OST_WAVE_BYTES;
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        const f32x16 *acc = which == 0 ? acc_dk : acc_dv;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 w;
                w[0] = tc_pack_bf16(acc[db][4 * g], acc[db][4 * g + 1]);
                w[1] = tc_pack_bf16(acc[db][4 * g + 2], acc[db][4 * g + 3]);
                const int d = 32 * db + 8 * g + 4 * h2;
                *reinterpret_cast<u32x2 *>(ost + kl * OST_LD + d * 2) = w;
            }
        __builtin_amdgcn_wave_barrier();
        const long col = (long)((which + 1) * p.E + h * 64) * 2;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int r = it * 8 + (lane >> 3), c = lane & 7;
            u32x4 w = *reinterpret_cast<const u32x4 *>(ost + r * OST_LD + c * 16);
            const int k = k0 + r;
            if (k < p.T) {
                if (which == 0 && p.rope_cos) w = rope_bwd8(w, p, k, c, 1.0f);
                *reinterpret_cast<u32x4 *>(p.dqkv + (row_base + k) * p.ldg_bytes + col + c * 16) = w;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------------
// f32 VALU kernels
// ------------------------------------------------------------------------------------------
constexpr int F32_RT = 32;   // rows per LDS tile

__global__ __launch_bounds__(64) void attn_bwd_dq_f32_kernel(const BwdParams p) {
    __shared__ __attribute__((aligned(16))) float ks[F32_RT][64];
    __shared__ __attribute__((aligned(16))) float vs[F32_RT][64];
    const int lane = threadIdx.x;
    const int qb = blockIdx.x % p.nblk;
    const int bh = blockIdx.x / p.nblk;
    const int h = bh % p.H, b = bh / p.H;
    const long row_base = (long)b * p.T;
    int qi = qb * 64 + lane;
    const bool q_ok = qi < p.T;
    qi = q_ok ? qi : p.T - 1;

    float q[64], g[64], dq[64];
    {
        const float *qs = reinterpret_cast<const float *>(p.qkv + (row_base + qi) * p.ldq_bytes) + h * 64;
        const float *gs = reinterpret_cast<const float *>(p.dout + (row_base + qi) * p.ldo_bytes) + h * 64;
#pragma unroll
        for (int d = 0; d < 64; d += 4) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(qs + d);
            const f32x4 c = *reinterpret_cast<const f32x4 *>(gs + d);
#pragma unroll
            for (int j = 0; j < 4; ++j) { q[d + j] = a[j]; g[d + j] = c[j]; dq[d + j] = 0.f; }
        }
    }
    const long si = ((long)b * p.H + h) * p.T + qi;
    const float lse = p.lse[si], dlt = p.delta[si];

    for (int k0 = 0; k0 < p.T; k0 += F32_RT) {
        const uint64_t word = key_word(p, b, k0 >> 6);
        const uint32_t bits = (uint32_t)(word >> (k0 & 63));
        if (bits == 0u) continue;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = i * 64 + lane;
            const int r = idx >> 4, c4 = idx & 15;
            int key = k0 + r;
            key = key < p.T ? key : p.T - 1;
            const float *rowp = reinterpret_cast<const float *>(p.qkv + (row_base + key) * p.ldq_bytes);
            *reinterpret_cast<f32x4 *>(&ks[r][c4 * 4]) = *reinterpret_cast<const f32x4 *>(rowp + p.E + h * 64 + c4 * 4);
            *reinterpret_cast<f32x4 *>(&vs[r][c4 * 4]) =
                *reinterpret_cast<const f32x4 *>(rowp + 2 * p.E + h * 64 + c4 * 4);
        }
        __syncthreads();
        for (int j = 0; j < F32_RT; ++j) {
            if (!((bits >> j) & 1u)) continue;
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < 64; ++d) {
                s = fmaf(q[d], ks[j][d], s);
                dp = fmaf(g[d], vs[j][d], dp);
            }
            const float ds = expf(s - lse) * (dp - dlt);
#pragma unroll
            for (int d = 0; d < 64; ++d) dq[d] = fmaf(ds, ks[j][d], dq[d]);
        }
    }
    if (q_ok) {
        float *dst = reinterpret_cast<float *>(p.dqkv + (row_base + qi) * p.ldg_bytes) + h * 64;
#pragma unroll
        for (int d = 0; d < 64; d += 4)
            *reinterpret_cast<f32x4 *>(dst + d) = f32x4{dq[d], dq[d + 1], dq[d + 2], dq[d + 3]};
    }
}

__global__ __launch_bounds__(64) void attn_bwd_dkv_f32_kernel(const BwdParams p) {
    __shared__ __attribute__((aligned(16))) float qs[F32_RT][64];
    __shared__ __attribute__((aligned(16))) float gs[F32_RT][64];
    __shared__ float lse_s[F32_RT], dlt_s[F32_RT];
    const int lane = threadIdx.x;
    const int kb = blockIdx.x % p.nblk;
    const int bh = blockIdx.x / p.nblk;
    const int h = bh % p.H, b = bh / p.H;
    const long row_base = (long)b * p.T;
    const long stat_base = ((long)b * p.H + h) * p.T;
    int ki = kb * 64 + lane;
    const bool in_range = ki < p.T;
    ki = in_range ? ki : p.T - 1;
    const bool k_ok = in_range && ((key_word(p, b, ki >> 6) >> (ki & 63)) & 1ull);

    float k[64], v[64], dk[64], dv[64];
    {
        const float *rowp = reinterpret_cast<const float *>(p.qkv + (row_base + ki) * p.ldq_bytes);
#pragma unroll
        for (int d = 0; d < 64; d += 4) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(rowp + p.E + h * 64 + d);
            const f32x4 c = *reinterpret_cast<const f32x4 *>(rowp + 2 * p.E + h * 64 + d);
#pragma unroll
            for (int j = 0; j < 4; ++j) { k[d + j] = a[j]; v[d + j] = c[j]; dk[d + j] = 0.f; dv[d + j] = 0.f; }
        }
    }

    for (int q0 = 0; q0 < p.T; q0 += F32_RT) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = i * 64 + lane;
            const int r = idx >> 4, c4 = idx & 15;
            int qi = q0 + r;
            qi = qi < p.T ? qi : p.T - 1;
            *reinterpret_cast<f32x4 *>(&qs[r][c4 * 4]) = *reinterpret_cast<const f32x4 *>(
                reinterpret_cast<const float *>(p.qkv + (row_base + qi) * p.ldq_bytes) + h * 64 + c4 * 4);
            *reinterpret_cast<f32x4 *>(&gs[r][c4 * 4]) = *reinterpret_cast<const f32x4 *>(
                reinterpret_cast<const float *>(p.dout + (row_base + qi) * p.ldo_bytes) + h * 64 + c4 * 4);
        }
        if (lane < F32_RT) {
            const int qi = q0 + lane;
            lse_s[lane] = qi < p.T ? p.lse[stat_base + qi] : INFINITY;
            dlt_s[lane] = qi < p.T ? p.delta[stat_base + qi] : 0.f;
        }
        __syncthreads();
        for (int j = 0; j < F32_RT; ++j) {
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < 64; ++d) {
                s = fmaf(qs[j][d], k[d], s);
                dp = fmaf(gs[j][d], v[d], dp);
            }
            const float pr = k_ok ? expf(s - lse_s[j]) : 0.f;
            const float ds = pr * (dp - dlt_s[j]);
#pragma unroll
            for (int d = 0; d < 64; ++d) {
                dv[d] = fmaf(pr, gs[j][d], dv[d]);
                dk[d] = fmaf(ds, qs[j][d], dk[d]);
            }
        }
    }
    if (in_range) {
        float *rowp = reinterpret_cast<float *>(p.dqkv + (row_base + ki) * p.ldg_bytes);
#pragma unroll
        for (int d = 0; d < 64; d += 4) {
            *reinterpret_cast<f32x4 *>(rowp + p.E + h * 64 + d) = f32x4{dk[d], dk[d + 1], dk[d + 2], dk[d + 3]};
            *reinterpret_cast<f32x4 *>(rowp + 2 * p.E + h * 64 + d) = f32x4{dv[d], dv[d + 1], dv[d + 2], dv[d + 3]};
        }
    }
}

int set_lds_attr(const void *fn, int bytes, const char *what) {
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        tc_set_error("%s: hipFuncSetAttribute: %s", what, hipGetErrorString(e));
        return TC_ELAUNCH;
    }
    return TC_OK;
}

}  // namespace

extern "C" int tc_attn_bwd(int dtype, const void *qkv, int ldq, const uint64_t *key_words, const void *out,
                           const void *d_out, const float *lse, float *delta, void *d_qkv, int ldg,
                           const float *rope_cos, const float *rope_sin, float q_scale, int B, int T, int H, int D,
                           void *stream) {
    TC_REQUIRE((rope_cos != nullptr) == (rope_sin != nullptr), "tc_attn_bwd: give both rope tables or neither");
    TC_REQUIRE(tc_aligned16(rope_cos) && tc_aligned16(rope_sin), "tc_attn_bwd: rope tables must be 16-byte aligned");
    TC_REQUIRE(dtype == TC_BF16 || dtype == TC_F32, "tc_attn_bwd: bad dtype %d", dtype);
    TC_REQUIRE(qkv && out && d_out && lse && delta && d_qkv, "tc_attn_bwd: null operand");
    TC_REQUIRE(D == 64, "tc_attn_bwd: head_dim %d unsupported (only 64)", D);
    TC_REQUIRE(B > 0 && T > 0 && H > 0, "tc_attn_bwd: bad shape B=%d T=%d H=%d", B, T, H);
    const int E = H * D;
    const int esz = dtype == TC_BF16 ? 2 : 4;
    TC_REQUIRE(ldq >= 3 * E && (ldq * esz) % 16 == 0, "tc_attn_bwd: ldq=%d too small or misaligned", ldq);
    TC_REQUIRE(ldg >= 3 * E && (ldg * esz) % 16 == 0, "tc_attn_bwd: ldg=%d too small or misaligned", ldg);
    TC_REQUIRE(tc_aligned16(qkv) && tc_aligned16(out) && tc_aligned16(d_out) && tc_aligned16(d_qkv),
               "tc_attn_bwd: operands must be 16-byte aligned");
    BwdParams p{};
    p.qkv = static_cast<const char *>(qkv);
    p.words = key_words;
    p.out = static_cast<const char *>(out);
    p.dout = static_cast<const char *>(d_out);
    p.lse = lse;
    p.delta = delta;
    p.dqkv = static_cast<char *>(d_qkv);
    p.rope_cos = rope_cos;
    p.rope_sin = rope_sin;
    p.q_scale = q_scale;
    p.ldq_bytes = (long)ldq * esz;
    p.ldo_bytes = (long)E * esz;
    p.ldg_bytes = (long)ldg * esz;
    p.B = B; p.T = T; p.H = H; p.E = E;
    p.nw = (T + 63) / 64;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // dP, S (twice), dQ, dK, dV products: 7 x 2 B H T^2 D
    TcProfScope prof(TC_PROF_ATTN, s, 14.0 * (double)B * H * (double)T * (double)T * D);

    const long rows = (long)B * T;
    const long dthreads = rows * H * 8;
    const long dgrid = (dthreads + 255) / 256;
    TC_REQUIRE(dgrid < (1L << 31), "tc_attn_bwd: grid too large");
    if (dtype == TC_BF16)
        TC_LAUNCH(delta_kernel<uint16_t>, dim3((unsigned)dgrid), dim3(256), 0, s,
                           static_cast<const uint16_t *>(out), static_cast<const uint16_t *>(d_out), delta, rows, T,
                           H);
    else
        TC_LAUNCH(delta_kernel<float>, dim3((unsigned)dgrid), dim3(256), 0, s,
                           static_cast<const float *>(out), static_cast<const float *>(d_out), delta, rows, T, H);
    TC_CHECK_LAUNCH("tc_attn_bwd(delta)");

    if (dtype == TC_BF16) {
        static bool attr_set = false;
        if (!attr_set) {
            int rc = set_lds_attr(reinterpret_cast<const void *>(attn_bwd_dq_bf16_kernel), DQ_LDS_BYTES, "tc_attn_bwd");
            if (rc != TC_OK) return rc;
            rc = set_lds_attr(reinterpret_cast<const void *>(attn_bwd_dkv_bf16_kernel), DKV_LDS_BYTES, "tc_attn_bwd");
            if (rc != TC_OK) return rc;
            attr_set = true;
        }
        p.nblk = (T + 127) / 128;
        const long grid = (long)p.nblk * H * B;
        TC_REQUIRE(grid < (1L << 31), "tc_attn_bwd: grid too large");
        TC_REQUIRE(p.nw <= kMaxWords, "tc_attn_bwd: T=%d exceeds the bf16 kernels' %d keys", T, kMaxWords * 64);
        TC_LAUNCH(attn_bwd_dq_bf16_kernel, dim3((unsigned)grid), dim3(256), DQ_LDS_BYTES, s, p);
        TC_CHECK_LAUNCH("tc_attn_bwd(dq)");
        TC_LAUNCH(attn_bwd_dkv_bf16_kernel, dim3((unsigned)grid), dim3(256), DKV_LDS_BYTES, s, p);
        TC_CHECK_LAUNCH("tc_attn_bwd(dkv)");
    } else {
        p.nblk = (T + 63) / 64;
        const long grid = (long)p.nblk * H * B;
        TC_REQUIRE(grid < (1L << 31), "tc_attn_bwd: grid too large");
        TC_LAUNCH(attn_bwd_dq_f32_kernel, dim3((unsigned)grid), dim3(64), 0, s, p);
        TC_CHECK_LAUNCH("tc_attn_bwd(dq)");
        TC_LAUNCH(attn_bwd_dkv_f32_kernel, dim3((unsigned)grid), dim3(64), 0, s, p);
        TC_CHECK_LAUNCH("tc_attn_bwd(dkv)");
        if (rope_cos)     // the exactness path undoes the rotation with the standalone kernel
            return tc_rope_bwd(dtype, d_qkv, ldg, rope_cos, rope_sin, B * T, T, E, D, q_scale, stream);
    }
    return TC_OK;
}

// ---- allocation-free variant ------------------------------------------------------------------
// tc_attn_bwd needs one scratch buffer beyond its inputs and outputs: delta[b,h,t] (f32).  Callers
// that cannot allocate device memory (code under HIP graph capture, tight serving loops) get it from
// this file-scope array instead: kStaticDeltaFloats floats (16 MiB) that every GPU the library runs
// on holds until the process ends.  `scratch` is only a host-side stand-in for the symbol; the
// device address comes from hipGetSymbolAddress, once per device.
namespace {
constexpr size_t kStaticDeltaFloats = (size_t)4 << 20;
constexpr int kMaxDevices = 64;
float *g_static_delta[kMaxDevices] = {};
}  // namespace
static __device__ __align__(256) float scratch[kStaticDeltaFloats];

// Device address of `scratch` on the current device, resolved once per device.
static int static_delta(const char *who, float **out) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) {
        tc_set_error("%s: cannot identify the current device", who);
        return TC_ELAUNCH;
    }
    float *delta = __atomic_load_n(&g_static_delta[dev], __ATOMIC_ACQUIRE);
    if (!delta) {
        // first call on this device: every GPU has its own copy of the array at its own address
        void *sym = nullptr;
        const hipError_t e = hipGetSymbolAddress(&sym, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !sym) {
            tc_set_error("%s: hipGetSymbolAddress: %s", who, hipGetErrorString(e));
            return TC_ELAUNCH;
        }
        delta = static_cast<float *>(sym);
        __atomic_store_n(&g_static_delta[dev], delta, __ATOMIC_RELEASE);   // concurrent first calls store the same value
    }
    *out = delta;
    return TC_OK;
}

extern "C" int tc_init_device(void) {
    float *delta = nullptr;
    return static_delta("tc_init_device", &delta);
}

extern "C" int tc_attn_bwd_static(int dtype, const void *qkv, int ldq, const uint64_t *key_words, const void *out,
                                  const void *d_out, const float *lse, void *d_qkv, int ldg, const float *rope_cos,
                                  const float *rope_sin, float q_scale, int B, int T, int H, int D, void *stream) {
    TC_REQUIRE(B > 0 && T > 0 && H > 0, "tc_attn_bwd_static: bad shape B=%d T=%d H=%d", B, T, H);
    TC_REQUIRE((size_t)B * (size_t)H * (size_t)T <= kStaticDeltaFloats,
               "tc_attn_bwd_static: B*H*T = %zu exceeds the built-in scratch of %zu floats; use tc_attn_bwd with a "
               "caller-supplied delta buffer",
               (size_t)B * H * T, kStaticDeltaFloats);
    float *delta = nullptr;
    const int rc = static_delta("tc_attn_bwd_static", &delta);
    if (rc != TC_OK) return rc;
    return tc_attn_bwd(dtype, qkv, ldq, key_words, out, d_out, lse, delta, d_qkv, ldg, rope_cos, rope_sin, q_scale, B, T,
                       H, D, stream);
}
