// Weight-gradient GEMM on row-major operands (gfx950, bf16):
//   C[I,J] = sum_m A[m,I] * B[m,J]          A = dY [M, lda], B = X [M, ldb], C f32
// The reduction runs over the slow (token) dimension of both operands, so neither has the
// "k contiguous" layout the matrix cores want.  Tiles of 64 tokens x 128 columns are staged
// row-major with direct global->LDS loads and the fragments are read with the transposing LDS
// read (ds_read_b64_tr_b16), which hands every lane four consecutive-token values of its
// column: no transposed copy of dY or X is ever written to memory.
// Workgroup: 256 threads, 128x128 output tile, 2x2 waves of 64x64 (four 32x32x16 MFMA blocks).
// grid.y splits the token range; split s writes its partial tile to out + s*slab (f32), to be
// summed by tc_reduce_slabs (deterministic, no atomics).
#include "common.h"
#include <type_traits>
#include <stdlib.h>

namespace {

constexpr int KT = 64;                         // tokens per stage
constexpr int TILE_BYTES = KT * 256;           // 64 rows x 128 bf16
constexpr int STAGE_BYTES = 2 * TILE_BYTES;    // A + B
constexpr int LDS_BYTES = 2 * STAGE_BYTES;

struct TnParams {
    const char *A, *B;
    float *out;
    long lda_bytes, ldb_bytes, slab_elems;
    int M, I, J, ldo, nk, splits, tiles_j;
};

// 16-byte chunk swizzle of a 256-byte row: the 8 rows one transposed read touches land on 8
// different 32-byte bank spans.
__device__ __forceinline__ int swz(int row) { return ((row & 3) << 2) | (((row >> 2) & 1) << 1); }

__global__ __launch_bounds__(256) void gemm_tn_kernel(const TnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wi = wid >> 1, wj = wid & 1;
    const int tile = tc_xcd_remap(blockIdx.x, gridDim.x);
    const int i0 = (tile / p.tiles_j) * 128, j0 = (tile % p.tiles_j) * 128;
    const int k_begin = (int)((long)p.nk * blockIdx.y / p.splits);
    const int k_end = (int)((long)p.nk * (blockIdx.y + 1) / p.splits);

    // staging: wave w loads rows [16w, 16w+16) of both tiles, 4 rows x 16 chunks per instruction
    const int srow = lane >> 4, schunk = lane & 15;
    long a_col[4], b_col[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = wid * 16 + i * 4 + srow;
        a_col[i] = (long)i0 * 2 + ((schunk ^ swz(r)) << 4);
        b_col[i] = (long)j0 * 2 + ((schunk ^ swz(r)) << 4);
    }
    auto stage = [&](int buf, int kt) {
        char *ab = smem + buf * STAGE_BYTES + wid * 16 * 256;
        char *bb = ab + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int m = kt * KT + wid * 16 + i * 4 + srow;
            m = m < p.M ? m : p.M - 1;
            __builtin_amdgcn_global_load_lds(TC_GLOBAL_PTR(p.A + m * p.lda_bytes + a_col[i]),
                                             TC_LDS_PTR(ab + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(TC_GLOBAL_PTR(p.B + m * p.ldb_bytes + b_col[i]),
                                             TC_LDS_PTR(bb + i * 1024), 16, 0, 0);
        }
    };

    // transposed-read offsets: lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3 of a
    // 4-row x 16-column block and receives the 4 rows of column (lane & 15)
    const int tq = (lane & 15) >> 2, tp = lane & 3, thalf = (lane >> 4) & 1, h2 = lane >> 5;
    int a_off[2][4][2], b_off[2][4][2];     // [block][k-step][lo/hi]
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int hi = 0; hi < 2; ++hi) {
                const int row = s * 16 + hi * 8 + 4 * h2 + tq;
                const int ca = wi * 8 + blk * 4 + thalf * 2 + (tp >> 1);
                const int cb = wj * 8 + blk * 4 + thalf * 2 + (tp >> 1);
                a_off[blk][s][hi] = row * 256 + ((ca ^ swz(row)) << 4) + 8 * (tp & 1);
                b_off[blk][s][hi] = TILE_BYTES + row * 256 + ((cb ^ swz(row)) << 4) + 8 * (tp & 1);
            }

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    auto frag = [&](const char *sb, int lo, int hi) {
        const bf16x4 l = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4 *)(sb + lo));
        const bf16x4 h = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4 *)(sb + hi));
        bf16x8 f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { f[j] = l[j]; f[4 + j] = h[j]; }
        return f;
    };

    if (k_begin < k_end) stage(0, k_begin);
    for (int kt = k_begin; kt < k_end; ++kt) {
        const int buf = (kt - k_begin) & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < k_end) stage(buf ^ 1, kt + 1);
        char *sb = smem + buf * STAGE_BYTES;
        const int valid = p.M - kt * KT;          // rows of this stage that exist
        if (valid < KT) {                          // last, partial stage: zero the missing rows of A
            for (int idx = threadIdx.x; idx < (KT - valid) * 16; idx += 256)
                *reinterpret_cast<u32x4 *>(sb + (valid + (idx >> 4)) * 256 + (idx & 15) * 16) = u32x4{0u, 0u, 0u, 0u};
            __syncthreads();
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8 af[2], bf[2];
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                af[blk] = frag(sb, a_off[blk][s][0], a_off[blk][s][1]);
                bf[blk] = frag(sb, b_off[blk][s][0], b_off[blk][s][1]);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
    }

    float *out = p.out + (long)blockIdx.y * p.slab_elems;
    const int jl = lane & 31;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int j = j0 + wj * 64 + b * 32 + jl;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = i0 + wi * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2;
                out[(long)i * p.ldo + j] = acc[a][b][r];
            }
        }
}

// ------------------------------------------------------------------------------------------
// 256 x 256 output tile, 8 waves (2 x 4, each 128 x 64 = 4 x 2 MFMA blocks): half the L2->LDS
// bytes per flop of the 128 x 128 kernel.  One workgroup per CU (128 KiB of LDS); the host
// picks the token split so that tiles x splits fills the chip once.
// ------------------------------------------------------------------------------------------
namespace big {
constexpr int TILE_BYTES = KT * 512;           // 64 rows x 256 bf16
constexpr int STAGE_BYTES = 2 * TILE_BYTES;
constexpr int LDS_BYTES = 2 * STAGE_BYTES;     // 128 KiB
}  // namespace big

__global__ __launch_bounds__(512) void gemm_tn256_kernel(const TnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wi = wid >> 2, wj = wid & 3;
    const int tile = blockIdx.x;
    const int i0 = (tile / p.tiles_j) * 256, j0 = (tile % p.tiles_j) * 256;
    const int k_begin = (int)((long)p.nk * blockIdx.y / p.splits);
    const int k_end = (int)((long)p.nk * (blockIdx.y + 1) / p.splits);

    // staging: wave w loads rows [8w, 8w+8) of both tiles, 2 rows x 32 chunks per instruction
    const int srow = lane >> 5, schunk = lane & 31;
    long a_col[4], b_col[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = wid * 8 + i * 2 + srow;
        a_col[i] = (long)i0 * 2 + ((schunk ^ swz(r)) << 4);
        b_col[i] = (long)j0 * 2 + ((schunk ^ swz(r)) << 4);
    }
    auto stage = [&](int buf, int kt) {
        char *ab = smem + buf * big::STAGE_BYTES + wid * 8 * 512;
        char *bb = ab + big::TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int m = kt * KT + wid * 8 + i * 2 + srow;
            m = m < p.M ? m : p.M - 1;
            __builtin_amdgcn_global_load_lds(TC_GLOBAL_PTR(p.A + m * p.lda_bytes + a_col[i]),
                                             TC_LDS_PTR(ab + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(TC_GLOBAL_PTR(p.B + m * p.ldb_bytes + b_col[i]),
                                             TC_LDS_PTR(bb + i * 1024), 16, 0, 0);
        }
    };

    // transposed reads: the row swizzle depends only on (row & 7) = 4*h2 + tq, a lane constant
    const int tq = (lane & 15) >> 2, tp = lane & 3, thalf = (lane >> 4) & 1, h2 = lane >> 5;
    const int row_l = 4 * h2 + tq;                        // + 16 s + 8 hi
    const int sw = swz(row_l);
    int a_base[4], b_base[2];
#pragma unroll
    for (int blk = 0; blk < 4; ++blk)
        a_base[blk] = row_l * 512 + (((wi * 16 + blk * 4 + thalf * 2 + (tp >> 1)) ^ sw) << 4) + 8 * (tp & 1);
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
        b_base[blk] = big::TILE_BYTES + row_l * 512 + (((wj * 8 + blk * 4 + thalf * 2 + (tp >> 1)) ^ sw) << 4) + 8 * (tp & 1);

    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    auto frag = [&](const char *sb, int off) {
        const bf16x4 l = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4 *)(sb + off));
        const bf16x4 h = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
            (__attribute__((address_space(3))) bf16x4 *)(sb + off + 8 * 512));
        bf16x8 f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { f[j] = l[j]; f[4 + j] = h[j]; }
        return f;
    };

    if (k_begin < k_end) stage(0, k_begin);
    for (int kt = k_begin; kt < k_end; ++kt) {
        const int buf = (kt - k_begin) & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < k_end) stage(buf ^ 1, kt + 1);
        char *sb = smem + buf * big::STAGE_BYTES;
        const int valid = p.M - kt * KT;
        if (valid < KT) {                          // last, partial stage: zero the missing rows of A
            for (int idx = threadIdx.x; idx < (KT - valid) * 32; idx += 512)
                *reinterpret_cast<u32x4 *>(sb + (valid + (idx >> 5)) * 512 + (idx & 31) * 16) = u32x4{0u, 0u, 0u, 0u};
            __syncthreads();
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8 bf[2];
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) bf[blk] = frag(sb, b_base[blk] + s * 16 * 512);
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const bf16x8 af = frag(sb, a_base[a] + s * 16 * 512);
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf[b], acc[a][b], 0, 0, 0);
            }
        }
    }

    float *out = p.out + (long)blockIdx.y * p.slab_elems;
    const int jl = lane & 31;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int j = j0 + wj * 64 + b * 32 + jl;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = i0 + wi * 128 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2;
                out[(long)i * p.ldo + j] = acc[a][b][r];
            }
        }
}

// ------------------------------------------------------------------------------------------
// 256 x 256 output tile, two wave groups in alternation (the schedule of gemm256x_kernel in
// gemm.hip): the two waves of a SIMD (row group 0 / 1) run half a phase apart, one issuing its
// 16 MFMAs of a phase (16x16x32: this shape sustains a higher clock than 32x32x16 on this chip)
// while the other issues that phase's transposing LDS reads and LDS-DMA loads.
// A k-tile (64 tokens) is four half-tiles of 64 rows x 256 bytes (the image of gemm_tn_kernel's
// tile): A0 A1 = columns i0 .. +127, +128 .. +255 of dY, B0 B1 likewise of X.  A wave owns a
// 64 (i) x 32 (j) block in each quarter of the output tile.  Phases of k-tile g (buffer g & 1):
//   0: read B0, A0   stage A1 of k-tile g+1   math quarter (0,0)
//   1: read B1       stage B0 of k-tile g+2   math (0,1)
//   2: read A1       stage A0 of k-tile g+2   math (1,1)
//   3: -             stage B1 of k-tile g+2   math (1,0); vmcnt(6) guarantees all of k-tile g+1
// Needs M % 64 == 0 (no partial stage to zero-fill) and at least two k-tiles per token split.
// ------------------------------------------------------------------------------------------
namespace alt {
constexpr int HALF_BYTES = KT * 256;            // 16 KiB
constexpr int DBUF_BYTES = 4 * HALF_BYTES;
constexpr int SLOT_A0 = 0, SLOT_A1 = HALF_BYTES, SLOT_B0 = 2 * HALF_BYTES, SLOT_B1 = 3 * HALF_BYTES;
constexpr int LDS_BYTES = 2 * DBUF_BYTES;       // 128 KiB
}  // namespace alt

// two k-steps (32 tokens each) of one 16-column block: 4 transposing reads at at + SLOT + s * 8192 (+ 4096)
template <int OFF>
__device__ __forceinline__ void tn_read1(unsigned at, bf16x4 &v) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(at), "n"(OFF));
}
template <int SLOT>
__device__ __forceinline__ void tn_read2(unsigned at, bf16x4 (&f)[2][2]) {
    tn_read1<SLOT>(at, f[0][0]); tn_read1<SLOT + 4096>(at, f[0][1]);
    tn_read1<SLOT + 8192>(at, f[1][0]); tn_read1<SLOT + 8192 + 4096>(at, f[1][1]);
}
// every LDS read of this wave has landed; the fragments are handed on through the statement so
// that no use of them can be scheduled above it
__device__ __forceinline__ void tn_wait(bf16x4 (&f)[2][2], bf16x4 (&g)[2][2]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(f[0][0]), "+v"(f[0][1]), "+v"(f[1][0]), "+v"(f[1][1]), "+v"(g[0][0]), "+v"(g[0][1]), "+v"(g[1][0]),
                   "+v"(g[1][1]));
}
__device__ __forceinline__ bf16x8 tn_join(bf16x4 l, bf16x4 h) {
    return __builtin_shufflevector(l, h, 0, 1, 2, 3, 4, 5, 6, 7);
}

#define TN_XBAR()                                  \
    do {                                           \
        __builtin_amdgcn_sched_barrier(0);         \
        __builtin_amdgcn_s_barrier();              \
        __builtin_amdgcn_sched_barrier(0);         \
    } while (0)

__global__ __launch_bounds__(512) void gemm_tn256x_kernel(const TnParams p) {
    using namespace alt;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int tile = blockIdx.x;
    const int i0 = (tile / p.tiles_j) * 256, j0 = (tile % p.tiles_j) * 256;
    const int k_begin = (int)((long)p.nk * blockIdx.y / p.splits);
    const int nkt = (int)((long)p.nk * (blockIdx.y + 1) / p.splits) - k_begin;      // >= 2 (host)

    // staging: a half-tile is 16 pieces of 4 rows x 256 B; this wave moves pieces 2 wid, 2 wid + 1
    const int srow = lane >> 4, schunk = lane & 15;
    unsigned a_src[2][2], b_src[2][2];           // [half][piece]: byte offset of this lane's 16 bytes in token row 0
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (wid * 2 + i) * 4 + srow;
        const unsigned rc = (unsigned)((schunk ^ swz(r)) << 4);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            a_src[h][i] = (unsigned)(r * p.lda_bytes) + (unsigned)(i0 + h * 128) * 2u + rc;
            b_src[h][i] = (unsigned)(r * p.ldb_bytes) + (unsigned)(j0 + h * 128) * 2u + rc;
        }
    }
    // buffer-descriptor loads: lane offset fixed, token offset scalar (the host keeps both operands below 4 GiB)
    const __amdgpu_buffer_rsrc_t rsrc_a = tc_buffer_rsrc(p.A), rsrc_b = tc_buffer_rsrc(p.B);
    auto stage_half = [&](int slot_off, const char *base, long ld_bytes, const unsigned (&src)[2], int kt) {
        char *dst = smem + slot_off + wid * 2048;
        const unsigned tok_off = (unsigned)((long)(k_begin + kt) * KT * ld_bytes);
        const __amdgpu_buffer_rsrc_t rsrc = base == p.A ? rsrc_a : rsrc_b;
        tc_blds16(rsrc, src[0], tok_off, dst);
        tc_blds16(rsrc, src[1], tok_off, dst + 1024);
    };

    // Transposed-read offsets inside a half-tile (row pitch 256 B).  A 16-lane group g = lane >> 4
    // addresses tokens 4g + q (q = (lane & 15) >> 2), bytes 8p .. 8p+7 (p = lane & 3) of a 16-column
    // block and receives, per lane, the 4 tokens of column lane & 15; a fragment of the 16x16x32
    // MFMA is that read plus the one 16 tokens further (its k order is the same for both operands);
    // + 8192 per k-step of 32 tokens.  The swizzle depends on (token & 7) = 4 (g & 1) + q only.
    const int tg = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3, c16 = lane & 15;
    const int row_l = 4 * tg + tq, sw = swz(row_l);
    int a_off[4], b_off[2];
#pragma unroll
    for (int blk = 0; blk < 4; ++blk)
        a_off[blk] = row_l * 256 + (((wr * 8 + blk * 2 + (tp >> 1)) ^ sw) << 4) + 8 * (tp & 1);
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
        b_off[blk] = row_l * 256 + (((wc * 4 + blk * 2 + (tp >> 1)) ^ sw) << 4) + 8 * (tp & 1);
    // The transposing reads are written in assembly: through the builtin the compiler cannot tell
    // them from the LDS-DMA writes in flight and puts vmcnt(0) in front of every group, which would
    // drain the prefetch each phase.  The price: nothing waits for them automatically, so every
    // fragment passes through wait_frags() (lgkmcnt(0)) before an MFMA reads it.
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char *)smem;
    unsigned a_addr[4], b_addr[2];
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) a_addr[blk] = lds0 + (unsigned)a_off[blk];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) b_addr[blk] = lds0 + (unsigned)b_off[blk];

    // ---- prologue: all of k-tile 0, and B0 A0 B1 of k-tile 1 ----
    stage_half(SLOT_B0, p.B, p.ldb_bytes, b_src[0], 0);
    stage_half(SLOT_A0, p.A, p.lda_bytes, a_src[0], 0);
    stage_half(SLOT_B1, p.B, p.ldb_bytes, b_src[1], 0);
    stage_half(SLOT_A1, p.A, p.lda_bytes, a_src[1], 0);
    stage_half(DBUF_BYTES + SLOT_B0, p.B, p.ldb_bytes, b_src[0], 1);
    stage_half(DBUF_BYTES + SLOT_A0, p.A, p.lda_bytes, a_src[0], 1);
    stage_half(DBUF_BYTES + SLOT_B1, p.B, p.ldb_bytes, b_src[1], 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    TN_XBAR();
    if (wr == 1) TN_XBAR();                       // row group 1 runs half a phase behind

    f32x4 acc[8][4];                              // [mq * 4 + i block][nq * 2 + j block], D[i][j]
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int g = 0; g < nkt; ++g) {
        const int cur = (g & 1) * DBUF_BYTES, nxt = DBUF_BYTES - cur;
        bf16x4 bq[2][2][2][2];                    // [nq][j block][k-step][lower / upper 16 tokens]
        bf16x4 af[4][2][2];                       // [i block][k-step][lower / upper] of the current mq
        auto read_b = [&](auto nq_c) {
            constexpr int nq = decltype(nq_c)::value;
            tn_read2<(nq ? SLOT_B1 : SLOT_B0)>(b_addr[0] + cur, bq[nq][0]);
            tn_read2<(nq ? SLOT_B1 : SLOT_B0)>(b_addr[1] + cur, bq[nq][1]);
        };
        auto read_a = [&](auto mq_c, auto half_c) {          // i blocks 2 half, 2 half + 1
            constexpr int mq = decltype(mq_c)::value, hf = decltype(half_c)::value;
            tn_read2<(mq ? SLOT_A1 : SLOT_A0)>(a_addr[2 * hf] + cur, af[2 * hf]);
            tn_read2<(mq ? SLOT_A1 : SLOT_A0)>(a_addr[2 * hf + 1] + cur, af[2 * hf + 1]);
        };
        auto math = [&](int mq, int nq) {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                    for (int ib = 0; ib < 4; ++ib)
                        acc[mq * 4 + ib][nq * 2 + jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            tn_join(af[ib][s][0], af[ib][s][1]), tn_join(bq[nq][jb][s][0], bq[nq][jb][s][1]),
                            acc[mq * 4 + ib][nq * 2 + jb], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        // ---- phase 0 ----
        if (g + 1 < nkt) stage_half(nxt + SLOT_A1, p.A, p.lda_bytes, a_src[1], g + 1);
        read_b(I0{});
        read_a(I0{}, I0{});
        // the B0 reads (8 of the 16 issued; the counter holds 15): their slot is restaged in phase 1
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        read_a(I0{}, I1{});
        TN_XBAR();
        tn_wait(bq[0][0], bq[0][1]); tn_wait(af[0], af[1]); tn_wait(af[2], af[3]);
        math(0, 0);
        TN_XBAR();
        // ---- phase 1 ----
        const bool more = g + 2 < nkt;
        if (more) stage_half(cur + SLOT_B0, p.B, p.ldb_bytes, b_src[0], g + 2);
        read_b(I1{});
        TN_XBAR();
        tn_wait(bq[1][0], bq[1][1]);
        math(0, 1);
        TN_XBAR();
        // ---- phase 2 ----
        if (more) stage_half(cur + SLOT_A0, p.A, p.lda_bytes, a_src[0], g + 2);
        read_a(I1{}, I0{});
        read_a(I1{}, I1{});
        TN_XBAR();
        tn_wait(af[0], af[1]); tn_wait(af[2], af[3]);
        math(1, 1);
        TN_XBAR();
        // ---- phase 3 ----
        if (more) {
            stage_half(cur + SLOT_B1, p.B, p.ldb_bytes, b_src[1], g + 2);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        TN_XBAR();
        math(1, 0);
        TN_XBAR();
    }
    if (wr == 0) TN_XBAR();                       // pairs with row group 1's last barrier

    float *out = p.out + (long)blockIdx.y * p.slab_elems;
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int j = j0 + (b >> 1) * 128 + wc * 32 + (b & 1) * 16 + c16;
            const int i = i0 + (a >> 2) * 128 + wr * 64 + (a & 3) * 16 + 4 * tg;
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(long)(i + r) * p.ldo + j] = acc[a][b][r];
        }
}

// column sums (and optional bf16 copy) of a row-major matrix: 8 columns per thread
template <typename TI>
__global__ __launch_bounds__(256) void colsum_kernel(const TI *__restrict__ x, long ld, uint16_t *__restrict__ copy,
                                                     long ld_copy, float *__restrict__ colsum, int M, int N,
                                                     int rows_per_block) {
    __shared__ float red[32][65];
    const int cg = threadIdx.x & 7, rl = threadIdx.x >> 3;          // 8 column groups x 32 row lanes
    const int n = blockIdx.x * 64 + cg * 8;
    const int m_begin = blockIdx.y * rows_per_block;
    const int m_end = min(M, m_begin + rows_per_block);
    float s[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = 0.f;
    if (n < N) {
        for (int m = m_begin + rl; m < m_end; m += 32) {
            float v[8];
            if constexpr (sizeof(TI) == 2) {
                const u32x4 w = *reinterpret_cast<const u32x4 *>(x + m * ld + n);
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[2 * j] = tc_bf16_lo(w[j]); v[2 * j + 1] = tc_bf16_hi(w[j]); }
            } else {
                const f32x4 a = *reinterpret_cast<const f32x4 *>(x + m * ld + n);
                const f32x4 b = *reinterpret_cast<const f32x4 *>(x + m * ld + n + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
                if (copy) {
                    u32x4 w;
#pragma unroll
                    for (int j = 0; j < 4; ++j) w[j] = tc_pack_bf16(v[2 * j], v[2 * j + 1]);
                    *reinterpret_cast<u32x4 *>(copy + m * ld_copy + n) = w;
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] += v[j];
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[rl][cg * 8 + j] = s[j];
    __syncthreads();
    if (threadIdx.x < 64) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 32; ++r) t += red[r][threadIdx.x];
        const int col = blockIdx.x * 64 + threadIdx.x;
        if (colsum && col < N) atomicAdd(colsum + col, t);
    }
}

}  // namespace

extern "C" int tc_gemm_tn(const void *A, int lda, const void *B, int ldb, float *out, int ldo, int M, int I, int J,
                          int splits, void *stream) {
    TC_REQUIRE(A && B && out, "tc_gemm_tn: null operand");
    TC_REQUIRE(M > 0 && I > 0 && J > 0 && I % 128 == 0 && J % 128 == 0, "tc_gemm_tn: I=%d and J=%d must be multiples of 128",
               I, J);
    TC_REQUIRE(lda >= I && ldb >= J && ldo >= J && lda % 8 == 0 && ldb % 8 == 0, "tc_gemm_tn: bad leading dimensions");
    TC_REQUIRE(tc_aligned16(A) && tc_aligned16(B) && tc_aligned16(out), "tc_gemm_tn: operands must be 16-byte aligned");
    TC_REQUIRE(splits >= 1 && splits <= 256, "tc_gemm_tn: bad split count %d", splits);
    TnParams p{};
    p.A = static_cast<const char *>(A);
    p.B = static_cast<const char *>(B);
    p.out = out;
    p.lda_bytes = (long)lda * 2;
    p.ldb_bytes = (long)ldb * 2;
    p.slab_elems = (long)I * ldo;
    p.M = M; p.I = I; p.J = J; p.ldo = ldo;
    p.nk = (M + KT - 1) / KT;
    p.splits = splits;
    p.tiles_j = J / 128;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_tn_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_tn256_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, big::LDS_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_tn256x_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, alt::LDS_BYTES);
        if (e != hipSuccess) {
            tc_set_error("tc_gemm_tn: hipFuncSetAttribute: %s", hipGetErrorString(e));
            return TC_ELAUNCH;
        }
        attr_set = true;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    TcProfScope prof(TC_PROF_GEMM, s, 2.0 * (double)M * (double)I * (double)J);
    // the large tile when it divides the problem and the launch still covers most of the chip
    const long big_wgs = (long)(I / 256) * (J / 256) * splits;
    if (I % 256 == 0 && J % 256 == 0 && big_wgs >= 192) {
        p.tiles_j = J / 256;
        static int alt_on = -1;                   // TC_TN_ALT=0 keeps the single-group kernel
        if (alt_on < 0) { const char *e = getenv("TC_TN_ALT"); alt_on = e ? atoi(e) != 0 : 1; }
        // the alternating kernel has no partial stage and wants two k-tiles in every token split
        if (alt_on && M % KT == 0 && p.nk >= 2 * splits && (long)M * lda * 2 < (1L << 32) &&
            (long)M * ldb * 2 < (1L << 32))
            TC_LAUNCH(gemm_tn256x_kernel, dim3((I / 256) * (J / 256), splits), dim3(512), alt::LDS_BYTES, s, p);
        else
            TC_LAUNCH(gemm_tn256_kernel, dim3((I / 256) * (J / 256), splits), dim3(512), big::LDS_BYTES, s, p);
    } else {
        TC_LAUNCH(gemm_tn_kernel, dim3((I / 128) * (J / 128), splits), dim3(256), LDS_BYTES, s, p);
    }
    TC_CHECK_LAUNCH("tc_gemm_tn");
    return TC_OK;
}

extern "C" int tc_colsum(int dtype, const void *x, int ld, void *bf16_copy, int ld_copy, float *colsum, int M, int N,
                         void *stream) {
    TC_REQUIRE(dtype == TC_F32 || dtype == TC_BF16, "tc_colsum: bad dtype");
    TC_REQUIRE(x && (colsum || bf16_copy) && M > 0 && N > 0 && N % 8 == 0 && ld >= N && ld % 8 == 0,
               "tc_colsum: bad arguments (N and ld must be multiples of 8)");
    TC_REQUIRE(!bf16_copy || (dtype == TC_F32 && ld_copy >= N && ld_copy % 8 == 0),
               "tc_colsum: the bf16 copy is for f32 input");
    TC_REQUIRE(tc_aligned16(x) && tc_aligned16(bf16_copy), "tc_colsum: operands must be 16-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int rows_per_block = 512;
    const dim3 grid((N + 63) / 64, (M + rows_per_block - 1) / rows_per_block), block(256);
    if (dtype == TC_BF16)
        TC_LAUNCH(colsum_kernel<uint16_t>, grid, block, 0, s, static_cast<const uint16_t *>(x), (long)ld,
                           static_cast<uint16_t *>(nullptr), 0L, colsum, M, N, rows_per_block);
    else
        TC_LAUNCH(colsum_kernel<float>, grid, block, 0, s, static_cast<const float *>(x), (long)ld,
                           static_cast<uint16_t *>(bf16_copy), (long)ld_copy, colsum, M, N, rows_per_block);
    TC_CHECK_LAUNCH("tc_colsum");
    return TC_OK;
}
