// Fused multi-head self-attention forward for head_dim 64 (gfx950).
//
// bf16 kernel: a 512-thread workgroup (8 waves) owns 256 query rows of one (sequence,
// head); each wave owns 32 of them.  K/V tiles of 64 keys are staged global -> LDS with
// direct loads, double buffered (one 8-row piece of K and one of V per wave and tile: the
// issue of a direct-to-LDS load costs the wave about 60 cycles, so a tile shared by 8 waves
// is cheaper than one shared by 4).  Per tile and wave:
//   S^T = K . Q^T   (mfma_f32_32x32x16_bf16, A = K rows from LDS, B = Q kept in registers)
// leaves the query on the lane and 32 keys in registers, so the online softmax is
// lane-local plus one exchange with lane ^ 32;
//   O^T += V^T . P^T  takes the exponentiated accumulators directly as the B operand and
// reads V^T fragments with the transposing LDS read (ds_read_b64_tr_b16) from the
// row-major V tile.  The per-query rescale is then a per-lane scalar.
// Padded keys get -inf through 64-bit validity words (one per sequence and key tile).
//
// f32 kernels (exactness path): the same structure on mfma_f32_32x32x2_f32 (f32 operands, f32
// accumulation) for sequences of up to 8192 keys; beyond that one 64-lane wave per 64 queries, one
// query per lane, plain fmaf dot products over keys broadcast from LDS.
#include "common.h"
#include <type_traits>

namespace {

constexpr float kLog2e = 1.4426950408889634f;

struct AttnParams {
    const char *qkv;
    const uint64_t *words;
    char *out;
    float *lse;   // optional [B,H,T]: natural-log sum of exp of each query's scores (for backward)
    long ldq_bytes, ldo_bytes;
    int B, T, H, E;
    int nw;       // validity words per sequence = ceil(T/64)
    int nqb;      // query blocks per (b,h)
    // Packed (variable-length) batches: sequences lie one after the other in the row dimension and
    // blocks[i] = {first row of the sequence, its length, query block inside it, unused}; the grid
    // is one workgroup per (block, head), no key is padding, B / T / nqb / words are not used.
    const int4 *blocks;
};

// ------------------------------------------------------------------------------------------
// bf16
// ------------------------------------------------------------------------------------------
constexpr int KV_TILE_BYTES = 64 * 128;            // 64 keys x 64 bf16
constexpr int KV_STAGE_BYTES = 2 * KV_TILE_BYTES;  // K + V
constexpr int OST_LD = 144;                        // bytes per staged output row (128 + pad)
constexpr int OST_WAVE_BYTES = 32 * OST_LD;
#ifndef TC_ATTN_STAGES
#define TC_ATTN_STAGES 2
#endif
#ifndef TC_ATTN_WAVES
#define TC_ATTN_WAVES 8
#endif
constexpr int kWaves = TC_ATTN_WAVES;              // waves per workgroup, 32 queries each (4 or 8)
constexpr int kPieces = 8 / kWaves;                // 8-row pieces of K (and of V) a wave stages per tile
constexpr int kStages = TC_ATTN_STAGES;            // K/V ring depth: tiles in flight = kStages - 1
#ifdef TC_ABL_LDS
constexpr int ATTN_LDS_BYTES = TC_ABL_LDS;         // occupancy ablation
#else
constexpr int ATTN_LDS_BYTES = kStages * KV_STAGE_BYTES > kWaves * OST_WAVE_BYTES ? kStages * KV_STAGE_BYTES
                                                                                  : kWaves * OST_WAVE_BYTES;
#endif
//   // the output staging reuses the K/V stages after the loop

__device__ __forceinline__ int swz_k(int row) { return (row >> 1) & 7; }          // row reads
__device__ __forceinline__ int swz_v(int row) { return ((row >> 1) & 1) << 2; }   // transposed reads

constexpr int kMaxWords = 128;   // key tiles whose validity words are kept in LDS (T <= 8192)

__device__ __forceinline__ uint64_t key_word(const AttnParams &p, int b, int kt, int Tn, int nw) {
    if (p.words) return p.words[(long)b * nw + kt];
    const int rem = Tn - kt * 64;
    return rem >= 64 ? ~0ull : ((1ull << rem) - 1ull);
}

// Q2: the caller's q rows are multiplied by log2(e) / sqrt(head_dim) instead of 1 / sqrt(head_dim), so the
// MFMA results are the scores in base-2 units and the running offset enters as the C operand of the first
// MFMA of each chain: exp2 applies to the accumulators as they are (no multiply-add per score).
#ifndef TC_ATTN_REVERSE
#define TC_ATTN_REVERSE 1
#endif
template <bool Q2>
__global__ __launch_bounds__(64 * kWaves, 4) void attn_bf16_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int id = tc_xcd_remap(blockIdx.x, gridDim.x);
    int qb, h, b, Tn;
    long row_base;
    if (p.blocks) {
        const int4 blk = p.blocks[id / p.H];
        h = id % p.H;
        b = 0;
        row_base = blk.x;
        Tn = blk.y;
        qb = blk.z;
    } else {
        qb = id % p.nqb;
        const int bh = id / p.nqb;
        h = bh % p.H;
        b = bh / p.H;
#if TC_ATTN_REVERSE
        // last batch rows first: the q/k/v product before this kernel has just written them, and its tail is
        // what the Infinity Cache still holds (-2.4 % per launch in the forward's kernel trace)
        b = (int)(gridDim.x / (unsigned)(p.nqb * p.H)) - 1 - b;
#endif
        Tn = p.T;
        row_base = (long)b * p.T;
    }
    const int nw = (Tn + 63) >> 6;
    const int ql = lane & 31, h2 = lane >> 5;
    const int q0 = qb * (32 * kWaves) + wid * 32;
    const bool q_live = q0 < Tn;                   // wave-uniform

    // The key-validity words of this batch row go to LDS once: a global load consumed inside the
    // key loop would make the compiler wait for everything in flight, the K/V prefetch included.
    __shared__ uint64_t s_words[kMaxWords];
    __shared__ int s_danger;                       // Q2: some wave saw an exponential out of range
    for (int i = threadIdx.x; i < nw && i < kMaxWords; i += 64 * kWaves) s_words[i] = key_word(p, b, i, Tn, nw);
    if (threadIdx.x == 0) s_danger = 0;
    __syncthreads();
    auto tile_word = [&](int kt) { return s_words[kt]; };       // the host checks nw <= kMaxWords

    // number of key tiles up to the last one holding a valid key
    int nkt = nw;
    while (nkt > 0 && tile_word(nkt - 1) == 0ull) --nkt;
#ifdef TC_ABL_NKT
    nkt = nkt < TC_ABL_NKT ? nkt : TC_ABL_NKT;   // timing experiment: only the first key tiles
#endif

    // ---- Q fragments: B operand, lane (q, h2) holds Q[q][16 ks + 8 h2 .. +7] ----
    bf16x8 qf[4];
    {
        int qr = q0 + ql;
        qr = qr < Tn ? qr : Tn - 1;
        const char *src = p.qkv + (row_base + qr) * p.ldq_bytes + (long)h * 128 + h2 * 16;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8 *>(src + ks * 32);
    }

    // ---- staging: per tile and wave kPieces pieces of K and of V (8 rows x 128 B each) ----
    const int srow = lane >> 3;
    // Buffer-descriptor loads over this sequence's rows: the lane's row and column offset never
    // change, the key tile is a scalar offset.  Only a partial last tile builds per-lane offsets
    // (its rows past the sequence re-read row T-1; their keys are masked by the validity words).
    int s_row[2];
    unsigned k_col[2], v_col[2];  // byte offset of the lane's 16 bytes inside a row
#pragma unroll
    for (int i = 0; i < kPieces; ++i) {
        const int r = (wid * kPieces + i) * 8 + srow;
        s_row[i] = r;
        k_col[i] = (unsigned)(p.E + h * 64) * 2u + (unsigned)(((lane & 7) ^ swz_k(r)) << 4);
        v_col[i] = (unsigned)(2 * p.E + h * 64) * 2u + (unsigned)(((lane & 7) ^ swz_v(r)) << 4);
    }
    const __amdgpu_buffer_rsrc_t rsrc_kv = tc_buffer_rsrc(p.qkv + row_base * p.ldq_bytes);
    auto stage = [&](int buf, int kt) {
        char *kb = smem + buf * KV_STAGE_BYTES + wid * kPieces * 1024;
        char *vb = kb + KV_TILE_BYTES;
        if (kt * 64 + 64 <= Tn) {
            const unsigned tile_off = (unsigned)((long)kt * 64 * p.ldq_bytes);
#pragma unroll
            for (int i = 0; i < kPieces; ++i) {
                const unsigned row_off = (unsigned)(s_row[i] * p.ldq_bytes);
                tc_blds16(rsrc_kv, row_off + k_col[i], tile_off, kb + i * 1024);
                tc_blds16(rsrc_kv, row_off + v_col[i], tile_off, vb + i * 1024);
            }
        } else {
#pragma unroll
            for (int i = 0; i < kPieces; ++i) {
                int key = kt * 64 + s_row[i];
                key = key < Tn ? key : Tn - 1;
                const unsigned row_off = (unsigned)(key * p.ldq_bytes);
                tc_blds16(rsrc_kv, row_off + k_col[i], 0u, kb + i * 1024);
                tc_blds16(rsrc_kv, row_off + v_col[i], 0u, vb + i * 1024);
            }
        }
    };

    // ---- LDS read offsets ----
    // K rows: lane reads row kb*32 + ql, chunk 2 ks + h2; rows 32 apart share the swizzle, so the second
    // key block is the first one's address + 4096 (an immediate of the read)
    int k_off[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) k_off[ks] = ql * 128 + (((2 * ks + h2) ^ swz_k(ql)) << 4);
    // V transposed reads: lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3 of a
    // 4-key x 16-column block and receives column (lane & 15) of the 4 keys.
    const int tq = (lane & 15) >> 2, tp = lane & 3, thalf = (lane >> 4) & 1;
    auto v_off = [&](int db, int key_row) {      // key_row = first key of the block + tq
        const int c = 4 * db + 2 * thalf + (tp >> 1);
        return KV_TILE_BYTES + key_row * 128 + ((c ^ swz_v(key_row)) << 4) + 8 * (tp & 1);
    };

    f32x16 acc_o[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[db][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    // Q2: minus the running offset in all 16 registers of a score block's C operand (0 until a valid key is seen)
    f32x16 negm16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // LDS byte address of this lane's V reads for d block db, key rows 4 h2 + tq of a stage: the
    // key-block / k-step / upper-half terms are immediates of the read instruction
    const unsigned lds_base = (unsigned)(unsigned long)(__attribute__((address_space(3))) char *)smem;
    const unsigned v_lane[2] = {lds_base + (unsigned)v_off(0, 4 * h2 + tq), lds_base + (unsigned)v_off(1, 4 * h2 + tq)};
    auto v_read_pair = [&](unsigned at, int kb, int s, bf16x4 &lo, bf16x4 &hi) {
        // rows + (32 kb + 16 s) and + 8 more: the swizzle of a row depends on bit 1 of it only
        switch (kb * 2 + s) {
            case 0: asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:1024" : "=&v"(lo), "=&v"(hi) : "v"(at)); break;
            case 1: asm volatile("ds_read_b64_tr_b16 %0, %2 offset:2048\n\tds_read_b64_tr_b16 %1, %2 offset:3072" : "=&v"(lo), "=&v"(hi) : "v"(at)); break;
            case 2: asm volatile("ds_read_b64_tr_b16 %0, %2 offset:4096\n\tds_read_b64_tr_b16 %1, %2 offset:5120" : "=&v"(lo), "=&v"(hi) : "v"(at)); break;
            default: asm volatile("ds_read_b64_tr_b16 %0, %2 offset:6144\n\tds_read_b64_tr_b16 %1, %2 offset:7168" : "=&v"(lo), "=&v"(hi) : "v"(at)); break;
        }
    };

    // Q2 runs the key loop without tile maxima (see below) and watches for an exponential out of range; in that
    // case, which needs a score 2^40 above a query's first tile, the whole workgroup runs the loop a second time
    // with the maxima (the waves share the staged tiles and their barriers, so they repeat together).
    bool danger = false;
    auto key_loop = [&](auto with_max_c) {
    constexpr bool with_max = decltype(with_max_c)::value;
    if (nkt > 0) stage(0, 0);
    if (kStages == 3 && nkt > 1) stage(1, 1);
    int buf = 0, buf_next = kStages - 1;           // ring slots of tile kt and of the tile staged at kt
    for (int kt = 0; kt < nkt; ++kt) {
        // the 4 pieces of the younger tile in flight (3-deep ring) may stay outstanding
        if (kStages == 3 && kt + 1 < nkt) {
            if (kPieces == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifndef TC_ABL_NOBAR
        __syncthreads();
#endif
#ifndef TC_ABL_NOSTAGE
        if (kt + kStages - 1 < nkt) stage(buf_next, kt + kStages - 1);
#endif
        const char *sb = smem + buf * KV_STAGE_BYTES;
        const unsigned v_at[2] = {v_lane[0] + (unsigned)(buf * KV_STAGE_BYTES),
                                  v_lane[1] + (unsigned)(buf * KV_STAGE_BYTES)};
        buf_next = buf;
        buf = (buf + 1 == kStages) ? 0 : buf + 1;
        if (!q_live) continue;                     // a wave without queries only stages and synchronises

        // S^T[key][query]
        // the first MFMA of each chain takes a constant as its C operand: 0 (no accumulator clearing), or
        // minus the running offset (Q2)
        f32x16 acc_s[2];
        uint32_t p_pk[2][8];                       // Q2: P as bf16 pairs
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8 *>(sb + k_off[ks] + kb * 4096);
                acc_s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], ks == 0 ? (Q2 ? negm16 : zero16) : acc_s[kb], 0, 0, 0);
            }
        }

        // mask + online softmax in base 2; the padded-key select runs only for tiles that contain a
        // padded key (wave-uniform branch).
        const uint64_t word = tile_word(kt);
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
#ifndef TC_ABL_NOSOFTMAX
        // maximum of the tile's scores over the query's 64 keys (this lane's 32 and lane ^ 32's)
        auto tile_max = [&]() {
            // two chains of three-input maxima (v_max3_f32): 18 instructions instead of 25
            float tm0 = fmaxf(acc_s[0][0], acc_s[0][1]), tm1 = fmaxf(acc_s[1][0], acc_s[1][1]);
#pragma unroll
            for (int r = 2; r < 16; r += 2) {
                tm0 = fmaxf(fmaxf(tm0, acc_s[0][r]), acc_s[0][r + 1]);
                tm1 = fmaxf(fmaxf(tm1, acc_s[1][r]), acc_s[1][r + 1]);
            }
            // the exchange with lane ^ 32 stays in the vector ALU: no LDS round trip in the softmax's dependent
            // chain.  The swap leaves x[lane & 31] in a and x[32 + (lane & 31)] in b.
            float a = fmaxf(tm0, tm1), b2 = a;
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b2));
            return fmaxf(a, b2);
        };
        // Lazy running maximum: the offset only moves when a tile's maximum exceeds it by more than
        // kLazy (log2 units), so P stays below 2^kLazy and the rescale of O (32 multiplies per lane)
        // is skipped for nearly every tile.  Exact up to rounding: any per-query offset cancels in the
        // final normalisation, which is done in fp32.
        constexpr float kLazy = 8.0f;
        if constexpr (Q2) {
            // The scores already carry minus the offset, so the exponentials can be taken as they stand: no maximum
            // is needed once a query has its offset (every tile after its first valid key).  bf16 and f32 keep
            // their relative precision whatever the size of P, and what an offset too low by up to 2^40 loses of
            // earlier tiles is below f32 resolution; beyond that (`danger`) the loop is repeated with the maxima.
            if (with_max || __builtin_amdgcn_ballot_w64(m_run == -INFINITY) != 0ull) {
            // tmax is relative to the offset (absolute before the first valid key)
            const float tmax = tile_max();
            const bool first = m_run == -INFINITY;
            const bool move = first ? tmax != -INFINITY : tmax > kLazy;
            if (__builtin_amdgcn_ballot_w64(move) != 0ull) {             // rare: first tile, or a maximum 2^kLazy above the offset
                const float delta = move ? tmax : 0.f;                   // new offset - old offset
                const float alpha = (move && first) ? 0.f : __builtin_amdgcn_exp2f(-delta);
                if (move) m_run = first ? delta : m_run + delta;
                l_run *= alpha;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc_s[kb][r] -= delta;
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc_o[db][r] *= alpha;
                const float nm = m_run == -INFINITY ? 0.f : -m_run;
#pragma unroll
                for (int r = 0; r < 16; ++r) negm16[r] = nm;
            }
            }
            // P is packed to bf16 pairs at once (the form P.V multiplies) and the row sum is taken over those
            // rounded values: one dot-product instruction per pair instead of two adds, and numerator and
            // denominator of the softmax see the same P
            const bf16x2 ones = __builtin_bit_cast(bf16x2, 0x3F803F80u);
            float lsum0 = 0.f, lsum1 = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const float e0 = __builtin_amdgcn_exp2f(acc_s[kb][r]), e1 = __builtin_amdgcn_exp2f(acc_s[kb][r + 1]);
                    const uint32_t w = tc_pack_bf16(e0, e1);
                    p_pk[kb][r >> 1] = w;
                    if (kb == 0) lsum0 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, w), ones, lsum0, false);
                    else lsum1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, w), ones, lsum1, false);
                }
            const float ls = lsum0 + lsum1;
            l_run += ls;
            // ls >= 2^40, +inf or NaN, tested on the bit pattern: a sum of non-negative terms orders like its
            // unsigned image, every NaN lies above +inf there, and no floating-point compare is involved that
            // the file's no-NaN arithmetic flags could rewrite (a sign bit, which cannot occur, also counts)
            danger |= __builtin_bit_cast(uint32_t, ls) >= 0x53800000u;
        } else {
            const float tmax = tile_max() * kLog2e;
            const float m_new = (tmax > m_run + kLazy) ? tmax : m_run;     // m_run = -inf: the first valid tile sets it
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
            const f32x2 scale2 = {kLog2e, kLog2e}, neg_m2 = {-m_use, -m_use};
            f32x2 lsum2 = {0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2 t = f32x2{acc_s[kb][r], acc_s[kb][r + 1]} * scale2 + neg_m2;
#ifdef TC_ABL_NOEXP
                    const f32x2 e = t;
#else
                    const f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
#endif
                    acc_s[kb][r] = e[0];
                    acc_s[kb][r + 1] = e[1];
                    lsum2 += e;
                }
            l_run = l_run * alpha + (lsum2[0] + lsum2[1]);
            m_run = m_new;
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0ull) {      // skip when no lane's maximum moved
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc_o[db][r] *= alpha;
            }
        }

#endif
        // O^T[d][query] += V^T . P^T ; P fragment of k-step (kb, s) = registers 8s..8s+7
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8 pf;
                if constexpr (Q2) {
                    pf = __builtin_bit_cast(bf16x8, u32x4{p_pk[kb][4 * s], p_pk[kb][4 * s + 1], p_pk[kb][4 * s + 2], p_pk[kb][4 * s + 3]});
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) pf[j] = (__bf16)acc_s[kb][8 * s + j];
                }
                // The transposing reads are issued from assembly: through the builtin the compiler
                // cannot tell them from the LDS-DMA writes of the next tile in flight and waits for
                // that prefetch (vmcnt(0)) in the middle of every iteration.
#ifdef TC_ABL_NOPV
                acc_o[0][8 * s] += (float)pf[0] + (float)pf[2] + (float)pf[4] + (float)pf[6];
                continue;
#endif
                bf16x4 lo[2], hi[2];
                v_read_pair(v_at[0], kb, s, lo[0], hi[0]);
                v_read_pair(v_at[1], kb, s, lo[1], hi[1]);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0]), "+v"(hi[0]), "+v"(lo[1]), "+v"(hi[1]));
#pragma unroll
                for (int db = 0; db < 2; ++db)
                    acc_o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        __builtin_shufflevector(lo[db], hi[db], 0, 1, 2, 3, 4, 5, 6, 7), pf, acc_o[db], 0, 0, 0);
            }
    }
    };
    bool synced = false;                           // every wave is known to be done with the K/V stages
    if constexpr (Q2) {
        key_loop(std::false_type{});
        if (__builtin_amdgcn_ballot_w64(danger) != 0ull && lane == 0) s_danger = 1;
        __syncthreads();
        synced = true;
        if (s_danger != 0) {
            synced = false;
            // out of range somewhere in the workgroup: start again, this time with the tile maxima
            m_run = -INFINITY;
            l_run = 0.f;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc_o[db][r] = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) negm16[r] = 0.f;
            key_loop(std::true_type{});
        }
    } else {
        key_loop(std::true_type{});
    }

    // ---- epilogue: normalise, stage through the wave's LDS rows, store 16 B per lane ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    if (p.lse && h2 == 0 && q0 + ql < Tn)
        p.lse[((long)b * p.H + h) * p.T + q0 + ql] =
            l_tot > 0.f ? (m_run + __log2f(l_tot)) * 0.69314718055994531f : INFINITY;
    if (!synced) __syncthreads();          // every wave is done with the K/V stages
    char *ost = smem + wid * OST_WAVE_BYTES;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            u32x2 w;
            w[0] = tc_pack_bf16(acc_o[db][4 * g] * inv, acc_o[db][4 * g + 1] * inv);
            w[1] = tc_pack_bf16(acc_o[db][4 * g + 2] * inv, acc_o[db][4 * g + 3] * inv);
            const int d = 32 * db + 8 * g + 4 * h2;
            *reinterpret_cast<u32x2 *>(ost + ql * OST_LD + d * 2) = w;
        }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int r = it * 8 + (lane >> 3), c = lane & 7;
        const u32x4 w = *reinterpret_cast<const u32x4 *>(ost + r * OST_LD + c * 16);
        const int q = q0 + r;
        if (q < Tn)
            *reinterpret_cast<u32x4 *>(p.out + (row_base + q) * p.ldo_bytes + (long)h * 128 + c * 16) = w;
    }
}

// ------------------------------------------------------------------------------------------
// f32: one query per lane
// ------------------------------------------------------------------------------------------
constexpr int F32_KT = 32;   // keys per LDS tile

__global__ __launch_bounds__(64) void attn_f32_kernel(const AttnParams p) {
    __shared__ __attribute__((aligned(16))) float ks[F32_KT][64];
    __shared__ __attribute__((aligned(16))) float vs[F32_KT][64];
    const int lane = threadIdx.x;
    int qb, h, b, Tn;
    long row_base;
    if (p.blocks) {                                // packed batch: see AttnParams::blocks
        const int4 blk = p.blocks[blockIdx.x / p.H];
        h = blockIdx.x % p.H;
        b = 0;
        row_base = blk.x;
        Tn = blk.y;
        qb = blk.z;
    } else {
        qb = blockIdx.x % p.nqb;
        const int bh = blockIdx.x / p.nqb;
        h = bh % p.H;
        b = bh / p.H;
        Tn = p.T;
        row_base = (long)b * p.T;
    }
    const int nw = (Tn + 63) >> 6;
    int qi = qb * 64 + lane;
    const bool q_ok = qi < Tn;
    qi = q_ok ? qi : Tn - 1;

    float q[64], o[64];
    {
        const float *src = reinterpret_cast<const float *>(p.qkv + (row_base + qi) * p.ldq_bytes) + h * 64;
#pragma unroll
        for (int d = 0; d < 64; d += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(src + d);
            q[d] = v[0]; q[d + 1] = v[1]; q[d + 2] = v[2]; q[d + 3] = v[3];
        }
#pragma unroll
        for (int d = 0; d < 64; ++d) o[d] = 0.f;
    }
    float m_run = -INFINITY, l_run = 0.f;

    for (int k0 = 0; k0 < Tn; k0 += F32_KT) {
        const uint64_t word = key_word(p, b, k0 >> 6, Tn, nw);
        const uint32_t bits = (uint32_t)(word >> (k0 & 63));
        if (bits == 0u) continue;                                   // uniform: no valid key here
        __syncthreads();
        // 32 keys x 64 floats for K and V = 2 x 512 float4, 8 + 8 per lane
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = i * 64 + lane;        // float4 index in the tile
            const int r = idx >> 4, c4 = idx & 15;
            int key = k0 + r;
            key = key < Tn ? key : Tn - 1;
            const float *rowp = reinterpret_cast<const float *>(p.qkv + (row_base + key) * p.ldq_bytes);
            *reinterpret_cast<f32x4 *>(&ks[r][c4 * 4]) =
                *reinterpret_cast<const f32x4 *>(rowp + p.E + h * 64 + c4 * 4);
            *reinterpret_cast<f32x4 *>(&vs[r][c4 * 4]) =
                *reinterpret_cast<const f32x4 *>(rowp + 2 * p.E + h * 64 + c4 * 4);
        }
        __syncthreads();
        float s[F32_KT];
        float tmax = -INFINITY;
#pragma unroll
        for (int j = 0; j < F32_KT; ++j) {
            float acc = 0.f;
#pragma unroll
            for (int d = 0; d < 64; ++d) acc = fmaf(q[d], ks[j][d], acc);
            s[j] = ((bits >> j) & 1u) ? acc : -INFINITY;
            tmax = fmaxf(tmax, s[j]);
        }
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = expf(m_run - m_new);   // m_new is finite: the tile has a valid key
        float lsum = 0.f;
#pragma unroll
        for (int d = 0; d < 64; ++d) o[d] *= alpha;
#pragma unroll
        for (int j = 0; j < F32_KT; ++j) {
            const float e = expf(s[j] - m_new);
            lsum += e;
#pragma unroll
            for (int d = 0; d < 64; ++d) o[d] = fmaf(e, vs[j][d], o[d]);
        }
        l_run = l_run * alpha + lsum;
        m_run = m_new;
    }
    if (q_ok) {
        const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
        if (p.lse) p.lse[((long)b * p.H + h) * p.T + qi] = l_run > 0.f ? m_run + logf(l_run) : INFINITY;
        float *dst = reinterpret_cast<float *>(p.out + (row_base + qi) * p.ldo_bytes) + h * 64;
#pragma unroll
        for (int d = 0; d < 64; d += 4)
            *reinterpret_cast<f32x4 *>(dst + d) = f32x4{o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv};
    }
}

// ------------------------------------------------------------------------------------------
// f32 on the matrix unit: mfma_f32_32x32x2_f32 multiplies f32 operands exactly and accumulates in f32 (the
// rate of the f32 vector ALU, but 32 x 32 x 2 products per instruction instead of 64 per v_fma)
// ------------------------------------------------------------------------------------------
// Same structure as the bf16 kernel: a workgroup of 4 waves owns 128 queries of one (sequence, head), 32 per
// wave; K/V tiles of 64 keys (256-byte rows) come in by direct-to-LDS loads, double buffered.  Per tile and wave
//   S^T = K . Q^T : 2 key blocks x 32 MFMAs, step s multiplies head dimension s (lanes 0-31) and 32 + s (lanes
//         32-63); the lane's 32 query values stay in registers, K values are 4-byte LDS reads from rows whose
//         16-byte chunks are stored at chunk ^ (row & 7) (32 rows would otherwise share one bank);
//   O^T += V^T . P^T : register r of a score block holds keys k(r) (lanes 0-31) and k(r) + 4 (lanes 32-63), which
//         is the B operand as it stands; the A operand is V[that key][lane's column], a conflict-free row read.
// The softmax is the bf16 kernel's (base 2, lazy running maximum), all in f32.
namespace f32m {
constexpr int kWavesF = 4;
constexpr int ROW_BYTES = 256;
constexpr int TILE_BYTES = 64 * ROW_BYTES;         // 64 keys
constexpr int STAGE_BYTES = 2 * TILE_BYTES;        // K + V
constexpr int LDS_BYTES = 2 * STAGE_BYTES;         // two stages: 64 KiB
}  // namespace f32m

__global__ __launch_bounds__(64 * f32m::kWavesF) void attn_f32_mfma_kernel(const AttnParams p) {
    using namespace f32m;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int id = tc_xcd_remap(blockIdx.x, gridDim.x);
    int qb, h, b, Tn;
    long row_base;
    if (p.blocks) {
        const int4 blk = p.blocks[id / p.H];
        h = id % p.H;
        b = 0;
        row_base = blk.x;
        Tn = blk.y;
        qb = blk.z;
    } else {
        qb = id % p.nqb;
        const int bh = id / p.nqb;
        h = bh % p.H;
        b = bh / p.H;
        Tn = p.T;
        row_base = (long)b * p.T;
    }
    const int nw = (Tn + 63) >> 6;
    const int ql = lane & 31, h2 = lane >> 5;
    const int q0 = qb * (32 * kWavesF) + wid * 32;
    const bool q_live = q0 < Tn;                   // wave-uniform

    __shared__ uint64_t s_words[kMaxWords];
    for (int i = threadIdx.x; i < nw && i < kMaxWords; i += 64 * kWavesF) s_words[i] = key_word(p, b, i, Tn, nw);
    __syncthreads();
    int nkt = nw;
    while (nkt > 0 && s_words[nkt - 1] == 0ull) --nkt;

    // Q: lane (query ql, half h2) holds head dimensions 32 h2 .. 32 h2 + 31
    float qf[32];
    {
        int qr = q0 + ql;
        qr = qr < Tn ? qr : Tn - 1;
        const char *src = p.qkv + (row_base + qr) * p.ldq_bytes + (long)h * ROW_BYTES + h2 * 128;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(src + i * 16);
            qf[4 * i] = v[0]; qf[4 * i + 1] = v[1]; qf[4 * i + 2] = v[2]; qf[4 * i + 3] = v[3];
        }
    }

    // staging: a piece is 4 rows x 256 B (lane -> row lane >> 4, 16-byte chunk lane & 15); a wave moves 4 pieces
    // of K and 4 of V per tile.  K chunks are swizzled on the source side: LDS position c holds chunk c ^ (row & 7).
    const int srow = lane >> 4, spos = lane & 15;
    unsigned k_col[4], v_col;
    int s_row[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wid * 4 + i) * 4 + srow;
        s_row[i] = r;
        k_col[i] = (unsigned)(p.E + h * 64) * 4u + (unsigned)((spos ^ (r & 7)) << 4);
    }
    v_col = (unsigned)(2 * p.E + h * 64) * 4u + (unsigned)(spos << 4);
    const __amdgpu_buffer_rsrc_t rsrc_kv = tc_buffer_rsrc(p.qkv + row_base * p.ldq_bytes);
    auto stage = [&](int buf, int kt) {
        char *kb = smem + buf * STAGE_BYTES + wid * 4096;
        char *vb = kb + TILE_BYTES;
        const bool full = kt * 64 + 64 <= Tn;
        const unsigned tile_off = full ? (unsigned)((long)kt * 64 * p.ldq_bytes) : 0u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int key = s_row[i];
            if (!full) {                           // rows past the sequence re-read its last row (their keys are masked)
                key += kt * 64;
                key = key < Tn ? key : Tn - 1;
            }
            const unsigned row_off = (unsigned)(key * p.ldq_bytes);
            tc_blds16(rsrc_kv, row_off + k_col[i], tile_off, kb + i * 1024);
            tc_blds16(rsrc_kv, row_off + v_col, tile_off, vb + i * 1024);
        }
    };

    // LDS read offsets.  K: row ql (+ 32 per key block), head dimension s + 32 h2 -> chunk (s >> 2) + 8 h2, word s & 3
    int k_off[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) k_off[j] = ql * ROW_BYTES + (((j ^ (ql & 7)) + 8 * h2) << 4);
    // V: key k(r) + 4 h2 (+ 32 per key block), column ql (+ 32 per column block)
    const int v_off = TILE_BYTES + 4 * h2 * ROW_BYTES + 4 * ql;

    f32x16 acc_o[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[db][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    if (nkt > 0) stage(0, 0);
    for (int kt = 0; kt < nkt; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                           // tile kt landed; everyone is done with the other stage
        if (kt + 1 < nkt) stage((kt + 1) & 1, kt + 1);
        if (!q_live) continue;
        const char *sb = smem + (kt & 1) * STAGE_BYTES;

        f32x16 acc_s[2];
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                const float kv = *reinterpret_cast<const float *>(sb + k_off[s >> 2] + kb * 32 * ROW_BYTES + 4 * (s & 3));
                acc_s[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(kv, qf[s], s == 0 ? zero16 : acc_s[kb], 0, 0, 0);
            }

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
        float tm0 = fmaxf(acc_s[0][0], acc_s[0][1]), tm1 = fmaxf(acc_s[1][0], acc_s[1][1]);
#pragma unroll
        for (int r = 2; r < 16; r += 2) {
            tm0 = fmaxf(fmaxf(tm0, acc_s[0][r]), acc_s[0][r + 1]);
            tm1 = fmaxf(fmaxf(tm1, acc_s[1][r]), acc_s[1][r + 1]);
        }
        float tmax = fmaxf(tm0, tm1);
        {
            float a = tmax, b2 = tmax;             // maximum with lane ^ 32: the other 32 keys of the query
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b2));
            tmax = fmaxf(a, b2) * kLog2e;
        }
        constexpr float kLazy = 8.0f;              // lazy running maximum, as in the bf16 kernel
        const float m_new = (tmax > m_run + kLazy) ? tmax : m_run;
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
        float lsum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(acc_s[kb][r], kLog2e, -m_use));
                acc_s[kb][r] = e;
                lsum += e;
            }
        l_run = l_run * alpha + lsum;
        m_run = m_new;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0ull) {
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc_o[db][r] *= alpha;
        }

        // O^T[d][query] += V^T[d][key] P^T[key][query], one MFMA per score register and column block
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key_row = (r & 3) + 8 * (r >> 2) + 32 * kb;
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const float vv = *reinterpret_cast<const float *>(sb + v_off + key_row * ROW_BYTES + db * 128);
                    acc_o[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv, acc_s[kb][r], acc_o[db], 0, 0, 0);
                }
            }
    }

    // normalise and store: lane (query ql, half h2) holds columns 32 db + 8 g + 4 h2 .. + 3 in registers 4 g .. 4 g + 3
    float l_tot;
    {
        float a = l_run, b2 = l_run;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b2));
        l_tot = a + b2;
    }
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    const int q = q0 + ql;
    if (!q_live || q >= Tn) return;
    if (p.lse && h2 == 0)
        p.lse[((long)b * p.H + h) * p.T + q] = l_tot > 0.f ? (m_run + __log2f(l_tot)) * 0.69314718055994531f : INFINITY;
    char *dst = p.out + (row_base + q) * p.ldo_bytes + (long)h * ROW_BYTES;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4 *>(dst + (32 * db + 8 * g + 4 * h2) * 4) =
                f32x4{acc_o[db][4 * g] * inv, acc_o[db][4 * g + 1] * inv, acc_o[db][4 * g + 2] * inv, acc_o[db][4 * g + 3] * inv};
}

// ------------------------------------------------------------------------------------------
// pre-softmax scores (slow path for callers that ask for attention maps)
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void scores_kernel(const T *__restrict__ qkv, long ldq,
                                                     const uint8_t *__restrict__ pad,
                                                     float *__restrict__ scores, int Tn, int H,
                                                     int E) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    const int bh = blockIdx.z;
    const int h = bh % H, b = bh / H;
    if (j >= Tn) return;
    const T *q = qkv + ((long)b * Tn + i) * ldq + h * 64;
    const T *k = qkv + ((long)b * Tn + j) * ldq + E + h * 64;
    float acc = 0.f;
    for (int d = 0; d < 64; ++d) {
        float a, c;
        if constexpr (sizeof(T) == 2) {
            a = __builtin_bit_cast(float, (uint32_t)q[d] << 16);
            c = __builtin_bit_cast(float, (uint32_t)k[d] << 16);
        } else {
            a = q[d];
            c = k[d];
        }
        acc = fmaf(a, c, acc);
    }
    if (pad && pad[(long)b * Tn + j]) acc = -INFINITY;
    scores[(((long)b * H + h) * Tn + i) * Tn + j] = acc;
}

}  // namespace

static int attn_fwd_impl(int dtype, const void *qkv, int ldq, const uint64_t *key_words, void *out, float *lse,
                         int B, int T, int H, int D, bool q2, void *stream);
static int attn_fwd_packed_impl(int dtype, const void *qkv, int ldq, const int32_t *blocks, int n_blocks, int max_len,
                                void *out, int H, int D, double flops, bool q2, void *stream);

static int attn_f32_mfma_attr(const char *who) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(attn_f32_mfma_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, f32m::LDS_BYTES);
        if (e != hipSuccess) {
            tc_set_error("%s: hipFuncSetAttribute: %s", who, hipGetErrorString(e));
            return TC_ELAUNCH;
        }
        attr_set = true;
    }
    return TC_OK;
}

// The bf16 kernel's LDS request, once per variant
template <bool Q2>
static int attn_bf16_attr(const char *who) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(attn_bf16_kernel<Q2>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, ATTN_LDS_BYTES);
        if (e != hipSuccess) {
            tc_set_error("%s: hipFuncSetAttribute: %s", who, hipGetErrorString(e));
            return TC_ELAUNCH;
        }
        attr_set = true;
    }
    return TC_OK;
}

extern "C" int tc_attn_fwd(int dtype, const void *qkv, int ldq, const uint64_t *key_words,
                           void *out, int B, int T, int H, int D, void *stream) {
    return attn_fwd_impl(dtype, qkv, ldq, key_words, out, nullptr, B, T, H, D, false, stream);
}

extern "C" int tc_attn_fwd_q2(int dtype, const void *qkv, int ldq, const uint64_t *key_words, void *out, int B, int T,
                              int H, int D, void *stream) {
    TC_REQUIRE(dtype == TC_BF16, "tc_attn_fwd_q2: bf16 only (dtype %d)", dtype);
    return attn_fwd_impl(dtype, qkv, ldq, key_words, out, nullptr, B, T, H, D, true, stream);
}

extern "C" int tc_attn_fwd_train(int dtype, const void *qkv, int ldq, const uint64_t *key_words, void *out,
                                 float *lse, int B, int T, int H, int D, void *stream) {
    TC_REQUIRE(lse != nullptr, "tc_attn_fwd_train: lse is required");
    return attn_fwd_impl(dtype, qkv, ldq, key_words, out, lse, B, T, H, D, false, stream);
}

static int attn_fwd_impl(int dtype, const void *qkv, int ldq, const uint64_t *key_words, void *out, float *lse,
                         int B, int T, int H, int D, bool q2, void *stream) {
    TC_REQUIRE(dtype == TC_BF16 || dtype == TC_F32, "tc_attn_fwd: bad dtype %d", dtype);
    TC_REQUIRE(qkv && out, "tc_attn_fwd: null operand");
    TC_REQUIRE(D == 64, "tc_attn_fwd: head_dim %d unsupported (only 64)", D);
    TC_REQUIRE(B > 0 && T > 0 && H > 0, "tc_attn_fwd: bad shape B=%d T=%d H=%d", B, T, H);
    const int E = H * D;
    const int esz = dtype == TC_BF16 ? 2 : 4;
    TC_REQUIRE(ldq >= 3 * E && (ldq * esz) % 16 == 0, "tc_attn_fwd: ldq=%d too small or misaligned", ldq);
    TC_REQUIRE(tc_aligned16(qkv) && tc_aligned16(out), "tc_attn_fwd: operands must be 16-byte aligned");
    AttnParams p{};
    p.qkv = static_cast<const char *>(qkv);
    p.words = key_words;
    p.out = static_cast<char *>(out);
    p.lse = lse;
    p.ldq_bytes = (long)ldq * esz;
    p.ldo_bytes = (long)E * esz;
    p.B = B; p.T = T; p.H = H; p.E = E;
    p.nw = (T + 63) / 64;
    hipStream_t s = static_cast<hipStream_t>(stream);
    TcProfScope prof(TC_PROF_ATTN, s, 4.0 * (double)B * H * (double)T * (double)T * D);
    if (dtype == TC_BF16) {
        const int rc = q2 ? attn_bf16_attr<true>("tc_attn_fwd") : attn_bf16_attr<false>("tc_attn_fwd");
        if (rc != TC_OK) return rc;
        p.nqb = (T + 32 * kWaves - 1) / (32 * kWaves);
        const long grid = (long)p.nqb * H * B;
        TC_REQUIRE(grid < (1L << 31), "tc_attn_fwd: grid too large");
        TC_REQUIRE(p.nw <= kMaxWords, "tc_attn_fwd: T=%d exceeds the bf16 kernel's %d keys", T, kMaxWords * 64);
        if (q2) TC_LAUNCH(attn_bf16_kernel<true>, dim3((unsigned)grid), dim3(64 * kWaves), ATTN_LDS_BYTES, s, p);
        else TC_LAUNCH(attn_bf16_kernel<false>, dim3((unsigned)grid), dim3(64 * kWaves), ATTN_LDS_BYTES, s, p);
    } else if (p.nw <= kMaxWords) {
        const int rc = attn_f32_mfma_attr("tc_attn_fwd");
        if (rc != TC_OK) return rc;
        p.nqb = (T + 32 * f32m::kWavesF - 1) / (32 * f32m::kWavesF);
        const long grid = (long)p.nqb * H * B;
        TC_REQUIRE(grid < (1L << 31), "tc_attn_fwd: grid too large");
        TC_LAUNCH(attn_f32_mfma_kernel, dim3((unsigned)grid), dim3(64 * f32m::kWavesF), f32m::LDS_BYTES, s, p);
    } else {
        // beyond the validity-word table of the MFMA kernel: one query per lane, any length
        p.nqb = (T + 63) / 64;
        const long grid = (long)p.nqb * H * B;
        TC_REQUIRE(grid < (1L << 31), "tc_attn_fwd: grid too large");
        TC_LAUNCH(attn_f32_kernel, dim3((unsigned)grid), dim3(64), 0, s, p);
    }
    TC_CHECK_LAUNCH("tc_attn_fwd");
    return TC_OK;
}

// Packed batches: `blocks` (device, int32 x 4 per entry: first row of the sequence, its length, query
// block index, 0) lists the n_blocks (sequence, query block) pairs; query blocks are q_block_rows rows.
extern "C" int tc_attn_fwd_packed(int dtype, const void *qkv, int ldq, const int32_t *blocks, int n_blocks,
                                  int max_len, void *out, int H, int D, double flops, void *stream) {
    return attn_fwd_packed_impl(dtype, qkv, ldq, blocks, n_blocks, max_len, out, H, D, flops, false, stream);
}

extern "C" int tc_attn_fwd_packed_q2(int dtype, const void *qkv, int ldq, const int32_t *blocks, int n_blocks,
                                     int max_len, void *out, int H, int D, double flops, void *stream) {
    TC_REQUIRE(dtype == TC_BF16, "tc_attn_fwd_packed_q2: bf16 only (dtype %d)", dtype);
    return attn_fwd_packed_impl(dtype, qkv, ldq, blocks, n_blocks, max_len, out, H, D, flops, true, stream);
}

static int attn_fwd_packed_impl(int dtype, const void *qkv, int ldq, const int32_t *blocks, int n_blocks, int max_len,
                                void *out, int H, int D, double flops, bool q2, void *stream) {
    TC_REQUIRE(dtype == TC_BF16 || dtype == TC_F32, "tc_attn_fwd_packed: bad dtype %d", dtype);
    TC_REQUIRE(qkv && out && blocks, "tc_attn_fwd_packed: null operand");
    TC_REQUIRE(D == 64, "tc_attn_fwd_packed: head_dim %d unsupported (only 64)", D);
    TC_REQUIRE(n_blocks > 0 && H > 0 && max_len > 0, "tc_attn_fwd_packed: bad shape");
    const int E = H * D;
    const int esz = dtype == TC_BF16 ? 2 : 4;
    TC_REQUIRE(ldq >= 3 * E && (ldq * esz) % 16 == 0, "tc_attn_fwd_packed: ldq=%d too small or misaligned", ldq);
    TC_REQUIRE(tc_aligned16(qkv) && tc_aligned16(out) && tc_aligned16(blocks),
               "tc_attn_fwd_packed: operands must be 16-byte aligned");
    TC_REQUIRE((max_len + 63) / 64 <= kMaxWords,
               "tc_attn_fwd_packed: a sequence of %d tokens exceeds the kernels' %d keys", max_len, kMaxWords * 64);
    AttnParams p{};
    p.qkv = static_cast<const char *>(qkv);
    p.out = static_cast<char *>(out);
    p.ldq_bytes = (long)ldq * esz;
    p.ldo_bytes = (long)E * esz;
    p.H = H; p.E = E;
    p.blocks = reinterpret_cast<const int4 *>(blocks);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long grid = (long)n_blocks * H;
    TC_REQUIRE(grid < (1L << 31), "tc_attn_fwd_packed: grid too large");
    TcProfScope prof(TC_PROF_ATTN, s, flops);      // 4 * H * D * sum of T_s^2, known to the caller only
    if (dtype == TC_BF16) {
        const int rc = q2 ? attn_bf16_attr<true>("tc_attn_fwd_packed") : attn_bf16_attr<false>("tc_attn_fwd_packed");
        if (rc != TC_OK) return rc;
        if (q2) TC_LAUNCH(attn_bf16_kernel<true>, dim3((unsigned)grid), dim3(64 * kWaves), ATTN_LDS_BYTES, s, p);
        else TC_LAUNCH(attn_bf16_kernel<false>, dim3((unsigned)grid), dim3(64 * kWaves), ATTN_LDS_BYTES, s, p);
    } else {
        const int rc = attn_f32_mfma_attr("tc_attn_fwd_packed");
        if (rc != TC_OK) return rc;
        TC_LAUNCH(attn_f32_mfma_kernel, dim3((unsigned)grid), dim3(64 * f32m::kWavesF), f32m::LDS_BYTES, s, p);
    }
    TC_CHECK_LAUNCH("tc_attn_fwd_packed");
    return TC_OK;
}

// rows per query block of tc_attn_fwd_packed's block list for an element type
extern "C" int tc_attn_packed_block_rows(int dtype) { return dtype == TC_BF16 ? 32 * kWaves : 32 * f32m::kWavesF; }

extern "C" int tc_attn_scores(int dtype, const void *qkv, int ldq, const uint8_t *pad_flags,
                              float *scores, int B, int T, int H, int D, void *stream) {
    TC_REQUIRE(dtype == TC_BF16 || dtype == TC_F32, "tc_attn_scores: bad dtype %d", dtype);
    TC_REQUIRE(qkv && scores && D == 64 && B > 0 && T > 0 && H > 0, "tc_attn_scores: bad arguments");
    TC_REQUIRE(T <= 65535 && (long)B * H <= 65535, "tc_attn_scores: shape too large for the slow path");
    const int E = H * D;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((T + 255) / 256, T, B * H), block(256);
    if (dtype == TC_BF16)
        TC_LAUNCH(scores_kernel<uint16_t>, grid, block, 0, s, static_cast<const uint16_t *>(qkv),
                           (long)ldq, pad_flags, scores, T, H, E);
    else
        TC_LAUNCH(scores_kernel<float>, grid, block, 0, s, static_cast<const float *>(qkv),
                           (long)ldq, pad_flags, scores, T, H, E);
    TC_CHECK_LAUNCH("tc_attn_scores");
    return TC_OK;
}
