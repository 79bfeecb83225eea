// out[M,N] = epilogue(A[M,K] . W[N,K]^T + bias) on the gfx950 matrix cores.
//
// Both operands are K-contiguous, so a k-tile is 128 BYTES per row for either element type
// (64 bf16 or 32 f32); staging, swizzle and fragment addressing are written in bytes and
// shared by both.  MFMA: mfma_f32_16x16x32_bf16, or four mfma_f32_16x16x4_f32 per 16-byte
// fragment for f32 (exact f32 multiply-add chain).
//
// Staging: global -> LDS by direct loads (global_load_lds_dwordx4): one wave instruction
// writes 1 KiB = 8 rows x 128 B linearly.  Bank conflicts of the ds_read_b128 fragment
// reads are avoided by an XOR swizzle of the 16-byte chunk index with (row & 7), applied on
// the per-lane SOURCE address and again on the read address.  Two LDS stages: the loads of
// k-tile t+1 are in flight while tile t is multiplied.
//
// Two kernels share that machinery:
//   gemm256_kernel  256 x 256 tile, 8 waves (2 x 4, 128 x 64 each), one persistent workgroup
//                   per CU walking its share of the tiles; the first k-tile of the next tile
//                   is already in flight during the epilogue.  The per-CU global->LDS path
//                   moves half the bytes per flop of the small tile, which is what bounds
//                   these GEMMs (K = 768 is short, so L2->LDS traffic, not HBM, is the limit).
//   gemm128_kernel  128 x 128 tile, 4 waves (2 x 2, 64 x 64 each), two workgroups per CU;
//                   used for small M, for N not a multiple of 256.
//
// Epilogue: accumulators go through a wave-private LDS patch and come back row-wise, 8 or 16
// consecutive columns per lane, so that bias / GELU / residual / rotary arithmetic works on
// contiguous columns and global accesses are 16 bytes.
#include "common.h"
#include "gemm_params.h"
#include <type_traits>
#include <stdlib.h>

namespace {

using GemmParams = TcGemmParams;

#ifdef TC_GEMM_STAMPS
#define TC_STAMP(var) unsigned long long var = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F)
#else
#define TC_STAMP(var)
#endif

__device__ __forceinline__ void glds16(const char *src, char *lds_wave_base) {
    __builtin_amdgcn_global_load_lds(TC_GLOBAL_PTR(src), TC_LDS_PTR(lds_wave_base), 16, 0, 0);
}

// The large-tile kernel stages through a buffer descriptor instead (tc_blds16): lane offset fixed,
// k offset scalar, no vector add per piece; the K = 3072 product runs 7 % faster that way.
#ifndef TC_GEMM_BUFLOAD
#define TC_GEMM_BUFLOAD 1
#endif

// One k-step (16 bytes of K per row) of an MT x NT grid of 16x16 MFMA tiles.
template <typename T, int MT, int NT>
__device__ __forceinline__ void mma_step(const f32x4 (&a)[MT], const f32x4 (&b)[NT],
                                         f32x4 (&acc)[MT][NT]) {
    if constexpr (sizeof(T) == 2) {
        // the 16 bytes of a fragment are 8 bf16 = one 16x16x32 operand
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int ni = 0; ni < NT; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8, a[mi]), __builtin_bit_cast(bf16x8, b[ni]),
                    acc[mi][ni], 0, 0, 0);
    } else {
        // the 16 bytes are 4 consecutive k of one row; MFMA j of the group of four takes
        // element j from both operands: k-slot g of MFMA j stands for k = 4g + j in A and B
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                for (int ni = 0; ni < NT; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi][j], b[ni][j],
                                                                       acc[mi][ni], 0, 0, 0);
    }
}

template <typename T>
__device__ __forceinline__ void store8(char *out, long elem_index, const float (&v)[8]) {
    if constexpr (sizeof(T) == 2) {
        u32x4 w;
        w[0] = tc_pack_bf16(v[0], v[1]);
        w[1] = tc_pack_bf16(v[2], v[3]);
        w[2] = tc_pack_bf16(v[4], v[5]);
        w[3] = tc_pack_bf16(v[6], v[7]);
        *reinterpret_cast<u32x4 *>(out + elem_index * 2) = w;
    } else {
        f32x4 lo = {v[0], v[1], v[2], v[3]}, hi = {v[4], v[5], v[6], v[7]};
        *reinterpret_cast<f32x4 *>(out + elem_index * 4) = lo;
        *reinterpret_cast<f32x4 *>(out + elem_index * 4 + 16) = hi;
    }
}

// GELU for the bf16 path: erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below the
// bf16 rounding of the result), written so that negative arguments do not cancel:
//   gelu(x) = 0.5 x erfc(|x|/sqrt2) for x < 0,  0.5 x (2 - erfc(|x|/sqrt2)) otherwise.
__device__ __forceinline__ float gelu_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float poly = fmaf(t, 1.061405429f, -1.453152027f);
    poly = fmaf(t, poly, 1.421413741f);
    poly = fmaf(t, poly, -0.284496736f);
    poly = fmaf(t, poly, 0.254829592f);
    poly *= t;
    const float erfc_abs = poly * __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);
    return 0.5f * x * (x < 0.f ? erfc_abs : 2.0f - erfc_abs);
}

// Two values at a time: the polynomial and the scalings map onto packed f32 VALU ops.
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 x) {
    const f32x2 ax = {fabsf(x[0]), fabsf(x[1])};
    const f32x2 z = ax * 0.70710678118654752440f;
    const f32x2 d = z * 0.3275911f + 1.0f;
    const f32x2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    f32x2 poly = t * 1.061405429f + -1.453152027f;
    poly = t * poly + 1.421413741f;
    poly = t * poly + -0.284496736f;
    poly = t * poly + 0.254829592f;
    poly = poly * t;
    const f32x2 a = z * z * -1.4426950408889634f;
    const f32x2 e = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
    const f32x2 erfc_abs = poly * e;
    const f32x2 hx = x * 0.5f;
    f32x2 r;
    r[0] = hx[0] * (x[0] < 0.f ? erfc_abs[0] : 2.0f - erfc_abs[0]);
    r[1] = hx[1] * (x[1] < 0.f ? erfc_abs[1] : 2.0f - erfc_abs[1]);
    return r;
}

// GELU through a table of Phi in LDS (bf16 results of the large-tile kernel).  The pre-activation is
// rounded to bf16 first, as a bf16 linear layer followed by GELU does, so gelu(u) = u * Phi(u) is a
// function of 16 bits: entry (sign << 11) | clamp(|bits| - 0x3980, 0, 2047) covers 2^-12 <= |u| < 16
// exactly; below that Phi is 0.5 to 1e-4, above it 1 or 0.  Costs 13 VALU operations and two LDS
// gathers per pair of values instead of four transcendentals and about 17 operations.
constexpr int kPhiEntries = 4096;
__device__ __forceinline__ float phi_entry(int e) {
    const unsigned bits = (0x3980u + (e & 2047)) | ((unsigned)(e >> 11) << 15);
    const float u = __builtin_bit_cast(float, bits << 16);
    const float z = fabsf(u) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float poly = fmaf(t, 1.061405429f, -1.453152027f);
    poly = fmaf(t, poly, 1.421413741f);
    poly = fmaf(t, poly, -0.284496736f);
    poly = fmaf(t, poly, 0.254829592f);
    poly *= t;
    const float half_erfc = 0.5f * poly * __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);
    return u < 0.f ? half_erfc : 1.0f - half_erfc;
}
typedef __attribute__((ext_vector_type(2))) unsigned short u16x2;
// table entries of the two bf16 values packed in w
__device__ __forceinline__ f32x2 table_pair(uint32_t w, const char *tab) {
    const u16x2 mag = __builtin_bit_cast(u16x2, w & 0x7fff7fffu);
    u16x2 idx = __builtin_elementwise_sub_sat(mag, u16x2{0x3980, 0x3980});
    idx = __builtin_elementwise_min(idx, u16x2{2047, 2047});
    const uint32_t i2 = __builtin_bit_cast(uint32_t, idx) | ((w >> 4) & 0x08000800u);
    const float p0 = *reinterpret_cast<const float *>(tab + ((i2 << 2) & 0x3fffcu));
    const float p1 = *reinterpret_cast<const float *>(tab + ((i2 >> 14) & 0x3fffcu));
    return f32x2{p0, p1};
}
__device__ __forceinline__ f32x2 gelu_table2(f32x2 x, const char *phi) {
    const uint32_t w = tc_pack_bf16(x[0], x[1]);
    return f32x2{tc_bf16_lo(w), tc_bf16_hi(w)} * table_pair(w, phi);
}

// d/dx gelu(x) = Phi(x) + x phi(x); Phi from the same erfc as gelu_fast (bf16 path) or exact erf.
template <bool FAST>
__device__ __forceinline__ float gelu_grad(float x) {
    if constexpr (FAST) {
        // exp(-x^2/2) serves both the density and the erfc approximation
        const float z = fabsf(x) * 0.70710678118654752440f;
        const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
        float poly = fmaf(t, 1.061405429f, -1.453152027f);
        poly = fmaf(t, poly, 1.421413741f);
        poly = fmaf(t, poly, -0.284496736f);
        poly = fmaf(t, poly, 0.254829592f);
        const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);
        const float h = 0.5f * poly * t * e;
        const float cdf = x < 0.f ? h : 1.0f - h;
        return fmaf(x, 0.3989422804014327f * e, cdf);
    } else {
        const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
        const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
        return fmaf(x, pdf, cdf);
    }
}
template <typename T>
__device__ __forceinline__ float load_elem(const char *base, long idx) {
    if constexpr (sizeof(T) == 2)
        return __builtin_bit_cast(float, (uint32_t)(*reinterpret_cast<const uint16_t *>(base + idx * 2)) << 16);
    else
        return *reinterpret_cast<const float *>(base + idx * 4);
}

// Epilogue arithmetic on 8 consecutive columns n..n+7 of row m; c = A.W^T (no bias yet).
template <typename T, int EPI>
__device__ __forceinline__ void epilogue8(const GemmParams &p, int m, int n, int section,
                                          const f32x4 &c0, const f32x4 &c1, const float (&bias)[8]) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = c0[j] + bias[j]; v[4 + j] = c1[j] + bias[4 + j]; }
    const long o = (long)m * p.ldo + n;
    if constexpr (EPI == TC_EPI_BIAS) {
        store8<T>(p.out, o, v);
    } else if constexpr (EPI == TC_EPI_PLAIN_F32) {
        store8<float>(p.out + (long)blockIdx.y * p.slab_elems * 4, o, v);
    } else if constexpr (EPI == TC_EPI_BIAS_GELU || EPI == TC_EPI_BIAS_GELU_SAVE) {
        if constexpr (EPI == TC_EPI_BIAS_GELU_SAVE) store8<T>(p.out2, o, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = sizeof(T) == 2 ? gelu_fast(v[j]) : tc_gelu(v[j]);
        store8<T>(p.out, o, v);
    } else if constexpr (EPI == TC_EPI_DGELU) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= gelu_grad<sizeof(T) == 2>(load_elem<T>(p.aux, o + j));
        store8<T>(p.out, o, v);
    } else if constexpr (EPI == TC_EPI_BIAS_RESIDUAL) {
        const f32x4 r0 = *reinterpret_cast<const f32x4 *>(p.residual + o);
        const f32x4 r1 = *reinterpret_cast<const f32x4 *>(p.residual + o + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = r0[j] + v[j]; v[4 + j] = r1[j] + v[4 + j]; }
        store8<float>(p.out, o, v);
    } else {  // TC_EPI_QKV_ROPE: section 0 = q, 1 = k, 2 = v (uniform per workgroup)
        if (section == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] *= p.q_scale;
        }
        if (section < 2) {
            const int pos = p.pos ? p.pos[m] : m % p.T;
            const long t = (long)pos * (p.head_dim >> 1) + ((n % p.head_dim) >> 1);
            const f32x4 cs = *reinterpret_cast<const f32x4 *>(p.rope_cos + t);
            const f32x4 sn = *reinterpret_cast<const f32x4 *>(p.rope_sin + t);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x0 = v[2 * j], x1 = v[2 * j + 1];
                v[2 * j] = x0 * cs[j] - x1 * sn[j];
                v[2 * j + 1] = x1 * cs[j] + x0 * sn[j];
            }
        }
        store8<T>(p.out, o, v);
    }
}

__device__ __forceinline__ void load_bias8(const float *bias, int n, float (&b)[8]) {
    if (!bias) {
#pragma unroll
        for (int j = 0; j < 8; ++j) b[j] = 0.f;
        return;
    }
    const f32x4 b0 = *reinterpret_cast<const f32x4 *>(bias + n);
    const f32x4 b1 = *reinterpret_cast<const f32x4 *>(bias + n + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { b[j] = b0[j]; b[4 + j] = b1[j]; }
}

// =============================================================================================
// 128 x 128 tile, 4 waves
// =============================================================================================
namespace k128 {
constexpr int BM = 128, BN = 128;
constexpr int TILE_BYTES = 128 * 128;          // one operand k-tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;    // A + B
constexpr int CST_LD = 68;                     // f32 per staged C row (64 + 4 pad)
constexpr int CST_WAVE_BYTES = 64 * CST_LD * 4;
constexpr int LDS_BYTES = (4 * CST_WAVE_BYTES > 2 * STAGE_BYTES) ? 4 * CST_WAVE_BYTES : 2 * STAGE_BYTES;
}  // namespace k128

template <typename T, int EPI>
__global__ __launch_bounds__(256) void gemm128_kernel(const GemmParams p) {
    using namespace k128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = tc_xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- staging source addresses: 4 pieces of A and 4 of W per wave and stage ----
    const int srow = lane >> 3;                 // row inside an 8-row piece
    const int schunk = (lane & 7) ^ srow;       // logical 16-byte chunk this lane fetches
    const char *a_src[4], *b_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wid * 4 + i) * 8 + srow;
        int gm = m0 + r;
        gm = gm < p.M ? gm : p.M - 1;           // rows past M re-read the last row; never stored
        a_src[i] = p.A + (long)gm * p.lda_bytes + schunk * 16;
        b_src[i] = p.W + (long)(n0 + r) * p.ldw_bytes + schunk * 16;
    }
    auto stage = [&](int buf, int kt) {
        char *abase = smem + buf * STAGE_BYTES + wid * 4 * 1024;
        char *bbase = abase + TILE_BYTES;
        const long koff = (long)kt * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(a_src[i] + koff, abase + i * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(b_src[i] + koff, bbase + i * 1024);
    };

    // ---- fragment read offsets (bytes inside a stage) ----
    const int wm = wid >> 1, wn = wid & 1;
    const int frow = lane & 15, fgrp = lane >> 4;
    int a_off[2], b_off[2];                     // per k-step; + mi*16 rows added as immediates
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int c = ((ks * 4 + fgrp) ^ (frow & 7)) << 4;
        a_off[ks] = (wm * 64 + frow) * 128 + c;
        b_off[ks] = TILE_BYTES + (wn * 64 + frow) * 128 + c;
    }

    f32x4 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    // split-K: workgroup row blockIdx.y reduces k-tiles [k_begin, k_end) into its own f32 slab
    const int k_begin = (int)((long)p.nk * blockIdx.y / p.splits);
    const int k_end = (int)((long)p.nk * (blockIdx.y + 1) / p.splits);
    if (k_begin < k_end) stage(0, k_begin);
    for (int kt = k_begin; kt < k_end; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                         // tile kt landed; the other buffer is free
        if (kt + 1 < k_end) stage((kt - k_begin + 1) & 1, kt + 1);
        const char *sbase = smem + ((kt - k_begin) & 1) * STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f32x4 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = *reinterpret_cast<const f32x4 *>(sbase + a_off[ks] + i * 16 * 128);
                b[i] = *reinterpret_cast<const f32x4 *>(sbase + b_off[ks] + i * 16 * 128);
            }
            mma_step<T, 4, 4>(a, b, acc);
        }
    }

    // ---- epilogue ----
    __syncthreads();                             // every wave is done with the stage buffers
    float *cst = reinterpret_cast<float *>(smem) + wid * (64 * CST_LD);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                cst[(mi * 16 + fgrp * 4 + r) * CST_LD + ni * 16 + frow] = acc[mi][ni][r];
    __builtin_amdgcn_wave_barrier();

    const int rsub = lane >> 3, cg = lane & 7;
    const int n = n0 + wn * 64 + cg * 8;         // first of this lane's 8 columns
    float bias[8];
    load_bias8(p.bias, n, bias);
    const int section = EPI == TC_EPI_QKV_ROPE ? n0 / p.E : 0;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int row = it * 8 + rsub;
        const int m = m0 + wm * 64 + row;
        const f32x4 c0 = *reinterpret_cast<const f32x4 *>(cst + row * CST_LD + cg * 8);
        const f32x4 c1 = *reinterpret_cast<const f32x4 *>(cst + row * CST_LD + cg * 8 + 4);
        if (m < p.M) epilogue8<T, EPI>(p, m, n, section, c0, c1, bias);
    }
}

// =============================================================================================
// 256 x 256 tile, 8 waves, persistent
// =============================================================================================
namespace k256 {
constexpr int BM = 256, BN = 256;
constexpr int TILE_BYTES = 256 * 128;          // one operand k-tile (32 KiB)
constexpr int STAGE_BYTES = 2 * TILE_BYTES;    // A + B (64 KiB)
constexpr int LDS_BYTES = 2 * STAGE_BYTES;     // 128 KiB
}  // namespace k256

// Row / column offset of accumulator tile (mi, ni) inside the wave's part of the 256 x 256 tile.
// LAY 0: the wave owns one 128 x 64 block (rows mi*16, columns ni*16).  LAY 1: the wave owns a
// 64 x 32 block in each quarter of the tile (row blocks 128 apart, column blocks 128 apart), which
// is what lets one quarter of the accumulators depend on one half of each operand tile only.
// LAY 2: as LAY 1, with the W rows of a pair of accumulator tiles interleaved in groups of four
// (operand row i of tile nb is column (i >> 2) * 8 + nb * 4 + (i & 3) of the 32-column block), so
// that a lane's 4 + 4 values of the pair are 8 consecutive output columns: bf16 results leave as
// 16-byte stores, 64 contiguous bytes per output row and instruction instead of 32.
template <int LAY> __device__ __forceinline__ constexpr int mi_off(int mi) {
    return LAY ? (mi >> 2) * 128 + (mi & 3) * 16 : mi * 16;
}
template <int LAY> __device__ __forceinline__ constexpr int ni_off(int ni) {
    return LAY == 2 ? (ni >> 1) * 128 + (ni & 1) * 4 : LAY == 1 ? (ni >> 1) * 128 + (ni & 1) * 16 : ni * 16;
}

// Accumulators of one 128 x 64 wave tile start at bias (+ residual): no bias registers are held
// through the k-loop.  acc[ni][mi] = D[n][m] tiles of 16 x 16.
template <typename T, int EPI, int LAY = 0, int MI0 = 0, int MI1 = 8, int NI0 = 0, int NI1 = 4, bool SPLIT = false>
__device__ __forceinline__ void init_acc256(const GemmParams &p, f32x4 (&acc)[4][8], int mrow0, int ncol0) {
    constexpr bool kPreloadResidual = EPI == TC_EPI_BIAS_RESIDUAL && sizeof(T) == 2;
    if constexpr (SPLIT) {
        // residual stream as two bf16 planes: the lane's 8 consecutive columns of a pair of accumulator
        // tiles are one 16-byte load from each plane
        static_assert(kPreloadResidual && LAY == 2, "split residual planes: bf16 large tile, interleaved columns");
        const char *hi_base = reinterpret_cast<const char *>(p.residual);
#pragma unroll
        for (int ni = NI0; ni < NI1; ni += 2) {
            const f32x4 b0 = p.bias ? *reinterpret_cast<const f32x4 *>(p.bias + ncol0 + ni_off<LAY>(ni)) : f32x4{0.f, 0.f, 0.f, 0.f};
            const f32x4 b1 = p.bias ? *reinterpret_cast<const f32x4 *>(p.bias + ncol0 + ni_off<LAY>(ni + 1)) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int mi = MI0; mi < MI1; ++mi) {
                const int m = mrow0 + mi_off<LAY>(mi);
                const long at = ((long)(m < p.M ? m : p.M - 1) * p.ldo + ncol0 + ni_off<LAY>(ni)) * 2;
                const u32x4 h = *reinterpret_cast<const u32x4 *>(hi_base + at);
                const u32x4 l = *reinterpret_cast<const u32x4 *>(p.res_lo + at);
                acc[ni][mi] = f32x4{tc_bf16_lo(h[0]) + tc_bf16_lo(l[0]), tc_bf16_hi(h[0]) + tc_bf16_hi(l[0]),
                                    tc_bf16_lo(h[1]) + tc_bf16_lo(l[1]), tc_bf16_hi(h[1]) + tc_bf16_hi(l[1])} + b0;
                acc[ni + 1][mi] = f32x4{tc_bf16_lo(h[2]) + tc_bf16_lo(l[2]), tc_bf16_hi(h[2]) + tc_bf16_hi(l[2]),
                                        tc_bf16_lo(h[3]) + tc_bf16_lo(l[3]), tc_bf16_hi(h[3]) + tc_bf16_hi(l[3])} + b1;
            }
        }
    } else {
        f32x4 bias[4];
#pragma unroll
        for (int ni = NI0; ni < NI1; ++ni)
            bias[ni] = p.bias ? *reinterpret_cast<const f32x4 *>(p.bias + ncol0 + ni_off<LAY>(ni)) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mi = MI0; mi < MI1; ++mi) {
            if constexpr (kPreloadResidual) {
                const int m = mrow0 + mi_off<LAY>(mi);
                const float *rp = p.residual + (long)(m < p.M ? m : p.M - 1) * p.ldo + ncol0;
#pragma unroll
                for (int ni = NI0; ni < NI1; ++ni)
                    acc[ni][mi] = *reinterpret_cast<const f32x4 *>(rp + ni_off<LAY>(ni)) + bias[ni];
            } else {
#pragma unroll
                for (int ni = NI0; ni < NI1; ++ni) acc[ni][mi] = bias[ni];
            }
        }
    }

}

// Row sums for a folded LayerNorm: sum and sum of squares of a pair of bf16 values, one dot-product
// instruction each (products of bf16 values are exact in f32).
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ void row_sums(uint32_t pair, float &s1, float &s2) {
    const bf16x2_t w = __builtin_bit_cast(bf16x2_t, pair), ones = __builtin_bit_cast(bf16x2_t, 0x3F803F80u);
    s1 = __builtin_amdgcn_fdot2_f32_bf16(w, ones, s1, false);
    s2 = __builtin_amdgcn_fdot2_f32_bf16(w, w, s2, false);
}
// The four lanes that share a row (16 lanes apart) add up; lane group 0 leaves the f32 pair in the
// plane of this column wave (row = wave row group * 64 + row block offset + lane row).
__device__ __forceinline__ void put_row_sums(char *planes, int wm, int wn, int block_off, int frow, int fgrp, float s1,
                                             float s2) {
    s1 += __shfl_xor(s1, 16, 64);
    s2 += __shfl_xor(s2, 16, 64);
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    if (fgrp == 0) *reinterpret_cast<f32x2 *>(planes + (wn * 256 + wm * 64 + block_off + frow) * 8) = f32x2{s1, s2};
}

// Stores of the large-tile epilogues go through BOUNDED buffer descriptors (num_records = the bytes of the M valid
// rows): a row past M is dropped by the hardware, so no store sits behind a branch, every store instruction of a tile
// is always issued (the counted vmcnt waits rely on that) and the compiler is free to keep the lane exchanges of
// several row blocks in flight instead of waiting for each one before its store.  Offsets are 32-bit: the host
// keeps (M + 256) * ldo * element size below 4 GiB for these kernels.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tc_out_rsrc(const void *base, int rows, int ld, int esz) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)((unsigned)rows * (unsigned)ld * (unsigned)esz),
                                             0x00020000);
}
#ifndef TC_LOADA_AUX
#define TC_LOADA_AUX 0
#endif
#ifndef TC_LOADW_AUX
#define TC_LOADW_AUX 0
#endif
#ifndef TC_GEMM_REV
#define TC_GEMM_REV 1           // fc1 walks its row panels downwards (see kRev)
#endif
#ifndef TC_STORE_AUX
#define TC_STORE_AUX 0          // cache policy bits of the epilogue stores (2 = nt: streaming lines)
#endif
template <int AUX = TC_STORE_AUX>
__device__ __forceinline__ void tc_bstore16(__amdgpu_buffer_rsrc_t rsrc, u32x4 v, unsigned byte_off) {
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (int)byte_off, 0, AUX);
}
// Cache policy per result (0 = default, 2 = nt, streaming lines).  Measured inside the whole step (kernel traces of
// two builds on one box): streaming q/k/v rows, fc1 results or data-gradient results make their own product 5-15 %
// slower there, although the same product alone, launched back to back, gains 2-7 % from them.
#ifndef TC_BIAS_STORE_AUX
#define TC_BIAS_STORE_AUX 0
#endif
#ifndef TC_DGELU_STORE_AUX
#define TC_DGELU_STORE_AUX 0
#endif
#ifndef TC_SAVE_STORE_AUX
#define TC_SAVE_STORE_AUX 0      // GELU output of the training forward's fc1
#endif
#ifndef TC_QKV_STORE_AUX
#define TC_QKV_STORE_AUX 0
#endif
#ifndef TC_SAVE_U_STORE_AUX
#define TC_SAVE_U_STORE_AUX 0    // pre-activation kept for the backward pass
#endif
#ifndef TC_LO_STORE_AUX
#define TC_LO_STORE_AUX 0        // lo plane of the residual stream (read again two products later)
#endif

// Epilogue of one wave tile.  Phase 1: arithmetic in place (its table loads are not queued
// behind any store); phase 2: all stores.
template <typename T, int EPI, int LAY = 0, int MI0 = 0, int MI1 = 8, int NI0 = 0, int NI1 = 4, bool LDSX = false,
          bool SPLIT = false>
__device__ __forceinline__ void epilogue256(const GemmParams &p, f32x4 (&acc)[4][8], int tm, int tn, int mrow0,
                                            int ncol0, int wm, int wn, int frow, int fgrp,
                                            const char *bias_lds = nullptr, const char *rope_lds = nullptr,
                                            const char *step_lds = nullptr, const char *rows_lds = nullptr,
                                            char *stat_lds = nullptr) {
    // LDSX: rope_lds / step_lds hold the rotary slices (q/k/v epilogue), or rope_lds the Phi table
    // (GELU epilogues)
    using namespace k256;
    constexpr bool kPreloadResidual = EPI == TC_EPI_BIAS_RESIDUAL && sizeof(T) == 2;
    const int section = EPI == TC_EPI_QKV_ROPE ? (tn * BN) / p.E : 0;
    if (bias_lds) {
        // accumulators started at zero; the bias of the wave's 2 x 32 columns is in its LDS slot
        // (column half ni >> 1, then the lane's 8 columns, then the tile of the pair)
        if (rows_lds) {
            // LayerNorm folded in: acc = rstd_m * acc + ((-rstd_m mean_m) * s_n + c_n)
            int F = frow;
            asm volatile("" : "+v"(F));            // as above: no address of this block lives through the k-loop
#pragma unroll
            for (int ni = NI0; ni < NI1; ++ni) {
                const int o = ((ni >> 1) * 32 + fgrp * 8 + (ni & 1) * 4) * 4;
                const f32x4 c = *reinterpret_cast<const f32x4 *>(bias_lds + o);
                const f32x4 sn = *reinterpret_cast<const f32x4 *>(bias_lds + 256 + o);
#pragma unroll
                for (int mi = MI0; mi < MI1; ++mi) {
                    const f32x2 st = *reinterpret_cast<const f32x2 *>(rows_lds + ((mi >> 2) * 64 + (mi & 3) * 16 + F) * 8);
                    acc[ni][mi] = acc[ni][mi] * st[0] + (sn * st[1] + c);
                }
            }
        } else {
#pragma unroll
            for (int ni = NI0; ni < NI1; ++ni) {
                const f32x4 b = *reinterpret_cast<const f32x4 *>(
                    bias_lds + ((ni >> 1) * 32 + fgrp * 8 + (ni & 1) * 4) * 4);
#pragma unroll
                for (int mi = MI0; mi < MI1; ++mi) acc[ni][mi] += b;
            }
        }
    }
    if constexpr (EPI == TC_EPI_BIAS_GELU_SAVE) {
        // pre-activation goes out first (kept for the backward pass), then GELU in place
        if constexpr (LAY == 2) {
            int L = fgrp * 16 + frow;
        asm volatile("" : "+v"(L));                         // keeps the address arithmetic below out of the k-loop                  // lane exchange as for the final stores
            const int pull = (((L & 3) << 4) | (L >> 2)) << 2;
            const int mbase = tm * BM + wm * 64 + (L >> 2), nbase = tn * BN + wn * 32 + (L & 3) * 8;
            const __amdgpu_buffer_rsrc_t rs2 = tc_out_rsrc(p.out2, p.M, p.ldo, 2);
#pragma unroll
            for (int mi = MI0; mi < MI1; ++mi) {
                const int m = mbase + mi_off<LAY>(mi);
                const unsigned orow = (unsigned)m * (unsigned)p.ldo + (unsigned)nbase;
#pragma unroll
                for (int ni = NI0; ni < NI1; ni += 2) {
                    const f32x4 v = acc[ni][mi], v2 = acc[ni + 1][mi];
                    u32x4 w;
                    w[0] = tc_pack_bf16(v[0], v[1]);
                    w[1] = tc_pack_bf16(v[2], v[3]);
                    w[2] = tc_pack_bf16(v2[0], v2[1]);
                    w[3] = tc_pack_bf16(v2[2], v2[3]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[e] = (unsigned)__builtin_amdgcn_ds_bpermute(pull, (int)w[e]);
                    tc_bstore16<TC_SAVE_U_STORE_AUX>(rs2, w, (orow + ni_off<LAY>(ni)) * 2u);
                }
            }
        } else {
#pragma unroll
            for (int mi = MI0; mi < MI1; ++mi) {
                const int m = mrow0 + mi_off<LAY>(mi);
                if (m < p.M) {
                    const long orow = (long)m * p.ldo + ncol0;
#pragma unroll
                    for (int ni = NI0; ni < NI1; ++ni) {
                        const f32x4 v = acc[ni][mi];
                        u32x2 w;
                        w[0] = tc_pack_bf16(v[0], v[1]);
                        w[1] = tc_pack_bf16(v[2], v[3]);
                        *reinterpret_cast<u32x2 *>(p.out2 + (orow + ni_off<LAY>(ni)) * 2) = w;
                    }
                }
            }
        }
    }
    if constexpr (EPI == TC_EPI_DGELU) {
        // Column sums of the result (the bias gradient of the layer whose output gradient this is)
        // are taken on the way: rows in registers first, then the 16 lanes that hold the 16 rows of
        // a block, one atomic per column and wave.
        f32x4 csum[NI1 - NI0];
#pragma unroll
        for (int ni = NI0; ni < NI1; ++ni) csum[ni - NI0] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mi = MI0; mi < MI1; ++mi) {
            const int m = mrow0 + mi_off<LAY>(mi);
            const char *up = p.aux + ((long)(m < p.M ? m : p.M - 1) * p.ldo + ncol0) * 2;
            const float live = m < p.M ? 1.0f : 0.0f;
#pragma unroll
            for (int ni = NI0; ni < NI1; ++ni) {
                const u32x2 w = *reinterpret_cast<const u32x2 *>(up + ni_off<LAY>(ni) * 2);
                f32x4 v = acc[ni][mi];
                v[0] *= gelu_grad<true>(tc_bf16_lo(w[0]));
                v[1] *= gelu_grad<true>(tc_bf16_hi(w[0]));
                v[2] *= gelu_grad<true>(tc_bf16_lo(w[1]));
                v[3] *= gelu_grad<true>(tc_bf16_hi(w[1]));
                acc[ni][mi] = v;
                csum[ni - NI0] += v * live;
            }
        }
        if (p.colsum) {
#pragma unroll
            for (int ni = NI0; ni < NI1; ++ni) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // inclusive scan over the 16 lanes of a row by DPP shifts (VALU only): lane 15 ends with the sum
                    float t = csum[ni - NI0][e];
                    t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x111, 0xf, 0xf, true));
                    t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x112, 0xf, 0xf, true));
                    t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x114, 0xf, 0xf, true));
                    t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x118, 0xf, 0xf, true));
                    if (frow == 15) atomicAdd(p.colsum + ncol0 + ni_off<LAY>(ni) + e, t);
                }
            }
        }
    }
    if constexpr (EPI == TC_EPI_BIAS_GELU || EPI == TC_EPI_BIAS_GELU_SAVE) {
#pragma unroll
        for (int ni = NI0; ni < NI1; ++ni)
#pragma unroll
            for (int mi = MI0; mi < MI1; ++mi) {
                f32x4 v = acc[ni][mi];
                if constexpr (LDSX) {
                    const f32x2 lo = gelu_table2(f32x2{v[0], v[1]}, rope_lds), hi = gelu_table2(f32x2{v[2], v[3]}, rope_lds);
                    v = f32x4{lo[0], lo[1], hi[0], hi[1]};
                } else if constexpr (sizeof(T) == 2) {
                    const f32x2 lo = gelu_fast2(f32x2{v[0], v[1]}), hi = gelu_fast2(f32x2{v[2], v[3]});
                    v = f32x4{lo[0], lo[1], hi[0], hi[1]};
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = tc_gelu(v[e]);
                }
                acc[ni][mi] = v;
            }
    } else if constexpr (EPI == TC_EPI_QKV_ROPE) {
        if constexpr (LDSX) {
            static_assert(LAY == 2, "the LDS rotary path is written for the interleaved column order");
            // cos/sin of this wave's first 16 rows and the lane's 4 rotary pairs were staged into
            // LDS (cos: 1 KiB, lane-linear; sin: the next 1 KiB); step_lds holds rows 16 and 80 of
            // both tables (cos16, sin16, cos80, sin80: 32 pairs each).  Row blocks follow by angle
            // addition: +16 rows, and +80 rows from block 3 to block 4.
            if (section < 2) {
                const float qs = section == 0 ? p.q_scale : 1.0f;
                int L = fgrp * 16 + frow;
        asm volatile("" : "+v"(L));                         // keeps the address arithmetic below out of the k-loop
                f32x4 c = *reinterpret_cast<const f32x4 *>(rope_lds + L * 16);
                f32x4 sn = *reinterpret_cast<const f32x4 *>(rope_lds + 1024 + L * 16);
                const int p0 = (((ncol0 - tn * BN) & 63) >> 1) * 4;      // byte offset of the lane's first pair
                const f32x4 c16 = *reinterpret_cast<const f32x4 *>(step_lds + p0);
                const f32x4 s16 = *reinterpret_cast<const f32x4 *>(step_lds + 128 + p0);
#pragma unroll
                for (int mi = 0; mi < MI1; ++mi) {      // the recurrence starts at row block 0 whatever MI0 is
                    if (mi >= MI0) {
#pragma unroll
                        for (int ni = NI0; ni < NI1; ++ni) {
                            const int e = (ni & 1) * 2;
                            const f32x4 v = acc[ni][mi] * qs;
                            acc[ni][mi] = f32x4{v[0] * c[e] - v[1] * sn[e], v[1] * c[e] + v[0] * sn[e],
                                                v[2] * c[e + 1] - v[3] * sn[e + 1], v[3] * c[e + 1] + v[2] * sn[e + 1]};
                        }
                    }
                    if (mi + 1 < MI1) {
                        f32x4 cs = c16, ss = s16;
                        if (mi == 3) {
                            cs = *reinterpret_cast<const f32x4 *>(step_lds + 256 + p0);
                            ss = *reinterpret_cast<const f32x4 *>(step_lds + 384 + p0);
                        }
                        const f32x4 cn = c * cs - sn * ss;
                        sn = sn * cs + c * ss;
                        c = cn;
                    }
                }
            }
        } else if (section < 2) {
            const float qs = section == 0 ? p.q_scale : 1.0f;
            const int half = p.head_dim >> 1;
            // this lane's 4 columns of n-tile ni are rotary pairs pi(ni) + fgrp*2 + {0,1}
            const int tile_pos0 = p.pos ? 0 : (tm * BM) % p.T;
            const bool recur = sizeof(T) == 2 && !p.pos && tile_pos0 + BM <= p.T;
            // No sequence boundary inside the tile (recur): rows 16 apart differ by a fixed
            // angle, so only the first row's cos/sin are gathered (a gather touches 16 table
            // rows per load instruction, which is what makes it expensive) and the other seven
            // follow by angle addition with the table's row 16; 7 steps keep f32 accuracy.
            // One n-tile at a time keeps the live rotary state at 8 registers.
#pragma unroll
            for (int ni = NI0; ni < NI1; ++ni) {
                const int pi = ((ncol0 - tn * BN + ni_off<LAY>(ni)) % p.head_dim) >> 1;
                if (recur) {
                    const long row = (long)(tile_pos0 + (LAY ? wm * 64 : wm * 128) + mi_off<LAY>(MI0) + frow) * half + pi;
                    f32x2 c = *reinterpret_cast<const f32x2 *>(p.rope_cos + row);
                    f32x2 sn = *reinterpret_cast<const f32x2 *>(p.rope_sin + row);
                    const f32x2 c16 = *reinterpret_cast<const f32x2 *>(p.rope_cos + 16L * half + pi);
                    const f32x2 s16 = *reinterpret_cast<const f32x2 *>(p.rope_sin + 16L * half + pi);
                    // LAY 1: the step from row block 3 to row block 4 is 128 - 48 = 80 rows
                    constexpr bool kJump = LAY >= 1 && MI0 < 4 && MI1 > 4;
                    f32x2 c80 = c16, s80 = s16;
                    if constexpr (kJump) {
                        c80 = *reinterpret_cast<const f32x2 *>(p.rope_cos + 80L * half + pi);
                        s80 = *reinterpret_cast<const f32x2 *>(p.rope_sin + 80L * half + pi);
                    }
#pragma unroll
                    for (int mi = MI0; mi < MI1; ++mi) {
                        const f32x4 v = acc[ni][mi] * qs;
                        acc[ni][mi] = f32x4{v[0] * c[0] - v[1] * sn[0], v[1] * c[0] + v[0] * sn[0],
                                            v[2] * c[1] - v[3] * sn[1], v[3] * c[1] + v[2] * sn[1]};
                        const bool jump = kJump && mi == 3;
                        const f32x2 cs = jump ? c80 : c16, ss = jump ? s80 : s16;
                        const f32x2 cn = c * cs - sn * ss;
                        sn = sn * cs + c * ss;
                        c = cn;
                    }
                } else {
#pragma unroll
                    for (int mi = MI0; mi < MI1; ++mi) {
                        const int m = mrow0 + mi_off<LAY>(mi);
                        const int rpos = p.pos ? p.pos[m < p.M ? m : p.M - 1] : m % p.T;
                        const long row = (long)rpos * half + pi;
                        const f32x2 c = *reinterpret_cast<const f32x2 *>(p.rope_cos + row);
                        const f32x2 sn = *reinterpret_cast<const f32x2 *>(p.rope_sin + row);
                        const f32x4 v = acc[ni][mi] * qs;
                        acc[ni][mi] = f32x4{v[0] * c[0] - v[1] * sn[0], v[1] * c[0] + v[0] * sn[0],
                                            v[2] * c[1] - v[3] * sn[1], v[3] * c[1] + v[2] * sn[1]};
                    }
                }
            }
        }
    } else if constexpr (EPI == TC_EPI_BIAS_RESIDUAL && !kPreloadResidual) {
#pragma unroll
        for (int mi = MI0; mi < MI1; ++mi) {
            const int m = mrow0 + mi_off<LAY>(mi);
            const float *rp = p.residual + (long)(m < p.M ? m : p.M - 1) * p.ldo + ncol0;
#pragma unroll
            for (int ni = NI0; ni < NI1; ++ni)
                acc[ni][mi] = *reinterpret_cast<const f32x4 *>(rp + ni_off<LAY>(ni)) + acc[ni][mi];
        }
    }
    constexpr bool kF32Out = EPI == TC_EPI_BIAS_RESIDUAL || EPI == TC_EPI_PLAIN_F32 || sizeof(T) == 4;
#ifdef TC_GEMM_STAMPS
    if (p.nostore) return;
#endif
    if constexpr (LAY >= 1) {
        // In the accumulator layout the 16 lanes that are neighbours hold 16 different output
        // rows, so a store instruction would reach memory as 64 separate 16-byte writes.  The
        // 64 lanes exchange their 16 bytes first (lane 4 r + s takes the data of lane 16 s + r):
        // four neighbouring lanes then write 64 contiguous bytes of one row.
        int L = fgrp * 16 + frow;
        asm volatile("" : "+v"(L));                         // keeps the address arithmetic below out of the k-loop
        const int pull = (((L & 3) << 4) | (L >> 2)) << 2;
        const int trow = L >> 2, tseg = L & 3;
        const int mbase = tm * BM + wm * 64, nbase = tn * BN + wn * 32 + tseg * (LAY == 2 ? 8 : 4);
        auto exchange = [&](u32x4 w) {
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (unsigned)__builtin_amdgcn_ds_bpermute(pull, (int)w[e]);
            return o;
        };
        if constexpr (EPI == TC_EPI_BIAS_RESIDUAL && MI0 == 0 && MI1 == 8 && NI0 == 0 && NI1 == 4 && !SPLIT) {
            if (stat_lds) {
                // Row statistics for the LayerNorm folded into the next product: sum and sum of squares of
                // the bf16-rounded results (the values the next product multiplies).  Per wave: its 64
                // columns of 128 rows; the four lanes that share a row add up and lane group 0 leaves the
                // f32 pair in the plane of this column wave.  The kernel adds the four planes of a tile to
                // the global accumulators later (flush_row_sums), as integers.
#pragma unroll
                for (int mi = 0; mi < 8; ++mi) {
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) {
                        const f32x4 v = acc[ni][mi];
                        row_sums(tc_pack_bf16(v[0], v[1]), s1, s2);
                        row_sums(tc_pack_bf16(v[2], v[3]), s1, s2);
                    }
                    put_row_sums(stat_lds, wm, wn, mi_off<LAY>(mi), frow, fgrp, s1, s2);
                }
            }
        }
        // bounded descriptors: rows past p.M are dropped by the hardware (see tc_out_rsrc)
        const __amdgpu_buffer_rsrc_t rs_out = tc_out_rsrc(p.out, p.M, p.ldo, kF32Out && !SPLIT ? 4 : 2);
        const __amdgpu_buffer_rsrc_t rs_out2 = tc_out_rsrc(p.out2, p.out2 ? p.M : 0, p.ldo, 2);
        const __amdgpu_buffer_rsrc_t rs_lo = tc_out_rsrc(p.out_lo, SPLIT ? p.M : 0, p.ldo, 2);
#pragma unroll
        for (int mi = MI0; mi < MI1; ++mi) {
            const int m = mbase + mi_off<LAY>(mi) + trow;
            const unsigned orow = (unsigned)m * (unsigned)p.ldo + (unsigned)nbase;
            if constexpr (LAY == 2 && !kF32Out) {
#pragma unroll
                for (int ni = NI0; ni < NI1; ni += 2) {
                    const f32x4 v = acc[ni][mi], v2 = acc[ni + 1][mi];
                    u32x4 w;
                    w[0] = tc_pack_bf16(v[0], v[1]);
                    w[1] = tc_pack_bf16(v[2], v[3]);
                    w[2] = tc_pack_bf16(v2[0], v2[1]);
                    w[3] = tc_pack_bf16(v2[2], v2[3]);
                    w = exchange(w);
                    tc_bstore16<EPI == TC_EPI_QKV_ROPE ? TC_QKV_STORE_AUX : EPI == TC_EPI_BIAS_GELU_SAVE ? TC_SAVE_STORE_AUX : EPI == TC_EPI_DGELU ? TC_DGELU_STORE_AUX : EPI == TC_EPI_BIAS ? TC_BIAS_STORE_AUX : TC_STORE_AUX>(rs_out, w, (orow + ni_off<LAY>(ni)) * 2u);
                }
            } else if constexpr (SPLIT) {
                // result as two bf16 planes, hi = bf16(v), lo = bf16(v - hi): one 16-byte store each
                static_assert(LAY == 2 && EPI == TC_EPI_BIAS_RESIDUAL, "split planes: residual epilogue, interleaved columns");
                float s1 = 0.f, s2 = 0.f;               // row sums of the hi plane (this lane's own row, before the exchange)
#pragma unroll
                for (int ni = NI0; ni < NI1; ni += 2) {
                    const f32x4 v = acc[ni][mi], v2 = acc[ni + 1][mi];
                    u32x4 h, l;
                    h[0] = tc_pack_bf16(v[0], v[1]);
                    h[1] = tc_pack_bf16(v[2], v[3]);
                    h[2] = tc_pack_bf16(v2[0], v2[1]);
                    h[3] = tc_pack_bf16(v2[2], v2[3]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) row_sums(h[e], s1, s2);
                    l[0] = tc_pack_bf16(v[0] - tc_bf16_lo(h[0]), v[1] - tc_bf16_hi(h[0]));
                    l[1] = tc_pack_bf16(v[2] - tc_bf16_lo(h[1]), v[3] - tc_bf16_hi(h[1]));
                    l[2] = tc_pack_bf16(v2[0] - tc_bf16_lo(h[2]), v2[1] - tc_bf16_hi(h[2]));
                    l[3] = tc_pack_bf16(v2[2] - tc_bf16_lo(h[3]), v2[3] - tc_bf16_hi(h[3]));
                    h = exchange(h);
                    l = exchange(l);
                    tc_bstore16(rs_out2, h, (orow + ni_off<LAY>(ni)) * 2u);
                    tc_bstore16<TC_LO_STORE_AUX>(rs_lo, l, (orow + ni_off<LAY>(ni)) * 2u);
                }
                if (stat_lds) put_row_sums(stat_lds, wm, wn, mi_off<LAY>(mi), frow, fgrp, s1, s2);
            } else {
                static_assert(kF32Out, "bf16 results use the interleaved column order");
#pragma unroll
                for (int ni = NI0; ni < NI1; ni += (LAY == 2 ? 2 : 1)) {
                    const u32x4 w = exchange(__builtin_bit_cast(u32x4, acc[ni][mi]));
                    tc_bstore16(rs_out, w, (orow + ni_off<LAY>(ni)) * 4u);
                    if constexpr (LAY == 2) {
                        // interleaved column order: the pair's second tile is the lane's next 4 columns
                        const u32x4 w2 = exchange(__builtin_bit_cast(u32x4, acc[ni + 1][mi]));
                        tc_bstore16(rs_out, w2, (orow + ni_off<LAY>(ni + 1)) * 4u);
                        if constexpr (EPI == TC_EPI_BIAS_RESIDUAL) {
                            if (p.out2) {                         // bf16 copy for the folded LayerNorm's product
                                const f32x4 f = __builtin_bit_cast(f32x4, w), f2 = __builtin_bit_cast(f32x4, w2);
                                u32x4 h;
                                h[0] = tc_pack_bf16(f[0], f[1]);
                                h[1] = tc_pack_bf16(f[2], f[3]);
                                h[2] = tc_pack_bf16(f2[0], f2[1]);
                                h[3] = tc_pack_bf16(f2[2], f2[3]);
                                tc_bstore16(rs_out2, h, (orow + ni_off<LAY>(ni)) * 2u);
                            }
                        }
                    }
                }
            }
        }
    } else {
#pragma unroll
        for (int mi = MI0; mi < MI1; ++mi) {
            const int m = mrow0 + mi_off<LAY>(mi);
            if (m < p.M) {
                const long orow = (long)m * p.ldo + ncol0;
#pragma unroll
                for (int ni = NI0; ni < NI1; ++ni) {
                    const f32x4 v = acc[ni][mi];
                    if constexpr (kF32Out) {
                        *reinterpret_cast<f32x4 *>(p.out + (orow + ni_off<LAY>(ni)) * 4) = v;
                    } else {
                        u32x2 w;
                        w[0] = tc_pack_bf16(v[0], v[1]);
                        w[1] = tc_pack_bf16(v[2], v[3]);
                        *reinterpret_cast<u32x2 *>(p.out + (orow + ni_off<LAY>(ni)) * 2) = w;
                    }
                }
            }
        }
    }
}

// The MFMA operands are swapped here (A operand = W rows, B operand = activation rows), so an
// accumulator tile is D[n][m]: the lane holds 4 CONSECUTIVE output columns n of one output row m.
// The epilogue then needs no LDS: bias / GELU / rotary pairs / residual act on the lane's own 4
// columns and every store is 8 bytes (bf16) or 16 bytes (f32), 16 rows x 32..64 contiguous bytes
// per wave instruction.  For the bf16 residual epilogue the f32 residual tile is loaded straight
// into the accumulators before the k-loop (its latency hides behind the whole loop).
template <typename T, int EPI>
__global__ __launch_bounds__(512) void gemm256_kernel(const GemmParams p) {
    using namespace k256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wid >> 2, wn = wid & 3;                 // 2 x 4 waves, 128 (m) x 64 (n) each
    const int frow = lane & 15, fgrp = lane >> 4;
    constexpr bool kPreloadResidual = EPI == TC_EPI_BIAS_RESIDUAL && sizeof(T) == 2;

    // ---- this workgroup's tiles: the 8 workgroups-per-XCD classes (id & 7) own contiguous
    // tile ranges; inside a class, workgroup j takes tiles j, j + per_class, ... so that the
    // workgroups that run side by side on one XCD work on neighbouring tiles (shared A row
    // panels and W column panels hit that XCD's L2).
    const int ntiles = p.tiles_m * p.tiles_n;
    const int cls = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int per_class = (gridDim.x + 7 - cls) >> 3;      // workgroups in this class
    const int q = ntiles >> 3, r = ntiles & 7;
    const int first = cls < r ? cls * (q + 1) : r * (q + 1) + (cls - r) * q;
    const int count = q + (cls < r ? 1 : 0);
    const int my_count = count > j ? (count - j + per_class - 1) / per_class : 0;

    // tile index -> (row panel, column panel): groups of gm row panels, walked m-fastest
    // or, when gn < tiles_n, groups of gn column panels, each walked down all row panels with n
    // fastest: the group's W panels then stay in the XCD's L2 while the A row panels stream by
    auto tile_coords = [&](int tile, int &tm, int &tn) {
        if (p.gn < p.tiles_n) {
            const int per_group = p.gn * p.tiles_m;
            const int grp = tile / per_group, loc = tile - grp * per_group;
            const int cols_here = min(p.gn, p.tiles_n - grp * p.gn);
            tm = loc / cols_here;
            tn = grp * p.gn + (loc - tm * cols_here);
            return;
        }
        const int per_group = p.gm * p.tiles_n;
        const int grp = tile / per_group, loc = tile - grp * per_group;
        const int rows_here = min(p.gm, p.tiles_m - grp * p.gm);
        tn = loc / rows_here;
        tm = grp * p.gm + (loc - tn * rows_here);
    };

    // ---- staging: per stage and wave 4 pieces of A and 4 of W (8 rows x 128 B each) ----
    const int srow = lane >> 3;
    const int schunk = (lane & 7) ^ srow;
    unsigned a_src[4], b_src[4];                // byte offsets (the host keeps both operands < 4 GiB)
    auto set_load_tile = [&](int local) {
        int tm, tn;
        tile_coords(first + j + local * per_class, tm, tn);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (wid * 4 + i) * 8 + srow;
            int gm = tm * BM + row;
            gm = gm < p.M ? gm : p.M - 1;
            a_src[i] = (unsigned)gm * (unsigned)p.lda_bytes + schunk * 16;
            b_src[i] = (unsigned)(tn * BN + row) * (unsigned)p.ldw_bytes + schunk * 16;
        }
    };
    auto stage = [&](int buf, int kt) {
        char *abase = smem + buf * STAGE_BYTES + wid * 4 * 1024;
        char *bbase = abase + TILE_BYTES;
        const unsigned koff = (unsigned)kt * 128u;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(p.A + (a_src[i] + koff), abase + i * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(p.W + (b_src[i] + koff), bbase + i * 1024);
    };

    int a_off[2], b_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int c = ((ks * 4 + fgrp) ^ (frow & 7)) << 4;
        a_off[ks] = (wm * 128 + frow) * 128 + c;
        b_off[ks] = TILE_BYTES + (wn * 64 + frow) * 128 + c;
    }

    if (my_count == 0) return;
    int ld_tile = 0, ld_k = 0;                  // load cursor, one k-tile ahead of the math
    set_load_tile(0);
    stage(0, 0);
    ld_k = 1;
    if (ld_k == p.nk) { ld_k = 0; ld_tile = 1; if (ld_tile < my_count) set_load_tile(ld_tile); }

#ifdef TC_GEMM_STAMPS
    unsigned long long c_wait = 0, c_issue = 0, c_math = 0, c_epi = 0;
#endif
    int g = 0;                                   // running k-tile count = LDS stage parity
    bool prev_full = false;                      // previous tile had no row tail (all stores issued)
    for (int t = 0; t < my_count; ++t) {
        int tm, tn;
        tile_coords(first + j + t * per_class, tm, tn);
        const int mrow0 = tm * BM + wm * 128 + frow;         // + mi*16
        const int ncol0 = tn * BN + wn * 64 + fgrp * 4;      // + ni*16

        f32x4 acc[4][8];                                      // [ni][mi], D[n][m]
        init_acc256<T, EPI>(p, acc, mrow0, ncol0);

        for (int kt = 0; kt < p.nk; ++kt, ++g) {
            TC_STAMP(s0);
            // k-tile g was issued one iteration ago.  Right after a full tile's epilogue the
            // only younger operations are that epilogue's 32 stores and this tile's accumulator
            // initialisation loads; leaving those in flight keeps the store drain off the
            // critical path (the vector-memory counter retires in issue order).
            if (kt == 0 && t > 0 && prev_full) {
                if constexpr (kPreloadResidual) asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
                else if (p.bias) asm volatile("s_waitcnt vmcnt(36)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(32)" ::: "memory");      // no bias loads were issued
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();        // k-tile g landed; the other stage is free
            asm volatile("" ::: "memory");
            TC_STAMP(s1);
            if (ld_tile < my_count) {
                stage((g + 1) & 1, ld_k);
                if (++ld_k == p.nk) { ld_k = 0; if (++ld_tile < my_count) set_load_tile(ld_tile); }
            }
            TC_STAMP(s2);
            const char *sbase = smem + (g & 1) * STAGE_BYTES;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                f32x4 wb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    wb[i] = *reinterpret_cast<const f32x4 *>(sbase + b_off[ks] + i * 16 * 128);
#pragma unroll
                for (int hm = 0; hm < 2; ++hm) {          // 4 row tiles at a time: 16 MFMAs
                    f32x4 xa[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        xa[i] = *reinterpret_cast<const f32x4 *>(sbase + a_off[ks] + (hm * 4 + i) * 16 * 128);
                    if constexpr (sizeof(T) == 2) {
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                acc[ni][hm * 4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                    __builtin_bit_cast(bf16x8, wb[ni]), __builtin_bit_cast(bf16x8, xa[i]),
                                    acc[ni][hm * 4 + i], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                                for (int i = 0; i < 4; ++i)
                                    acc[ni][hm * 4 + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                        wb[ni][jj], xa[i][jj], acc[ni][hm * 4 + i], 0, 0, 0);
                    }
                }
            }
#ifdef TC_GEMM_STAMPS
            asm volatile("" :: "v"(acc[0][0]), "v"(acc[3][7]));
            TC_STAMP(s3);
            c_wait += s1 - s0; c_issue += s2 - s1; c_math += s3 - s2;
#endif
        }
        TC_STAMP(e0);

        // ---- epilogue of tile t (the next tile's first k-tile is already in flight) ----
        // phase 1: arithmetic in place (its table loads are not queued behind any store);
        // phase 2: all stores.
        prev_full = (tm * BM + BM) <= p.M;
        epilogue256<T, EPI>(p, acc, tm, tn, mrow0, ncol0, wm, wn, frow, fgrp);
#ifdef TC_GEMM_STAMPS
        TC_STAMP(e1);
        c_epi += e1 - e0;
#endif
    }
#ifdef TC_GEMM_STAMPS
    if (p.stamps && threadIdx.x == 0) {
        p.stamps[blockIdx.x * 4 + 0] = c_wait; p.stamps[blockIdx.x * 4 + 1] = c_issue;
        p.stamps[blockIdx.x * 4 + 2] = c_math; p.stamps[blockIdx.x * 4 + 3] = c_epi;
    }
#endif
}

// =============================================================================================
// 256 x 256 tile, 8 waves, persistent, two wave groups in alternation
// =============================================================================================
// Same tile and MFMA operand order as gemm256_kernel; what differs is the schedule.  The two waves
// that share a SIMD (wave w and w + 4: row group 0 and 1) run half a phase apart: while one issues
// its 16 MFMAs of a phase, the other issues that phase's LDS reads and LDS-DMA loads, so the matrix
// pipe never waits behind the same wave's memory instructions.
//
// A k-tile (64 k) is four half-tiles of 128 rows x 128 bytes: A0, A1 (tile rows 0..127, 128..255)
// and B0, B1 (tile columns likewise).  A wave owns a 64 x 32 block in each quarter of the output
// tile, so quarter (mq, nq) needs only A{mq} and B{nq}.  Phases of k-tile g (LDS buffer g & 1):
//   0: read B0 (4 x 16 B), A0 (8)   stage A1 of k-tile g+1      math quarter (0,0)
//   1: read B1 (4)                  stage B0 of k-tile g+2      math (0,1)
//   2: read A1 (8)                  stage A0 of k-tile g+2      math (1,1)
//   3: -                            stage B1 of k-tile g+2      math (1,0)
// Every slot is restaged at least two phases after its last read, except B0 (read in phase 0,
// restaged in phase 1): its four reads are issued first and retired (lgkmcnt(8)) before the
// phase's first barrier.  The vector-memory counter retires in issue order, so the one wait of a
// k-tile, vmcnt(6) in phase 3, leaves exactly the three half-tiles of k-tile g+2 in flight and
// guarantees all of k-tile g+1; the two barriers that follow it publish that to every wave.
namespace k256x {
constexpr int BM = 256, BN = 256;
constexpr int HALF_BYTES = 128 * 128;           // one half-tile (16 KiB)
constexpr int DBUF_BYTES = 4 * HALF_BYTES;      // A0 A1 B0 B1
constexpr int SLOT_A0 = 0, SLOT_A1 = HALF_BYTES, SLOT_B0 = 2 * HALF_BYTES, SLOT_B1 = 3 * HALF_BYTES;
// per wave, after the operand buffers: a 1 KiB slot with the bias of the wave's 2 x 32 columns (256 B)
// and, with a folded LayerNorm, the folded weight's row sums for the same columns (next 256 B); the
// rotary cos and sin of the wave's first 16 rows x the lane's 4 pairs (2 KiB); rows 16 and 80 of both
// tables (512 B)
constexpr int BIAS_OFF = 2 * DBUF_BYTES, ROPE_OFF = BIAS_OFF + 8 * 1024, STEP_OFF = ROPE_OFF + 8 * 2048;
// after those: (rstd, -rstd * mean) of the tile's 256 rows when a LayerNorm is folded in (one copy
// per row group of 128 rows; staged by column wave 0 of the group, published by the phase barriers)
constexpr int ROWS_OFF = STEP_OFF + 8 * 512;
constexpr int LDS_BYTES = ROWS_OFF + 2 * 1024;  // 158 KiB
}  // namespace k256x

// Adds a finished tile's row sums to the global accumulators: thread t takes (row t >> 1, statistic
// t & 1), converts the four column waves' f32 partials to integers in units of 2^-24 (an exact image
// of each) and adds their sum with one 64-bit atomic.  Integer adds: the result does not depend on
// the order in which the column tiles of a row arrive.
template <bool kFresh = false>
__device__ __forceinline__ void flush_row_sums(const GemmParams &p, const char *planes, int tm) {
    int t = threadIdx.x;
    unsigned long long *acc_base = p.row_acc;
    if constexpr (kFresh) {
        // called from inside a k-loop whose registers are all taken: the thread index and the accumulator
        // address are made here, from the lane counter and from scalar registers, instead of living through
        // the loop (where they were spilled, and every reload waited for the staging loads in flight)
        t = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * 64 + __builtin_amdgcn_mbcnt_hi(-1, __builtin_amdgcn_mbcnt_lo(-1, 0));
        asm volatile("" : "+v"(t));
        asm volatile("" : "+s"(acc_base));
    }
    const int row = t >> 1, which = t & 1;
    long long sum = 0;
#pragma unroll
    for (int pl = 0; pl < 4; ++pl)
        sum += (long long)(*reinterpret_cast<const float *>(planes + (pl * 256 + row) * 8 + which * 4) * 16777216.f);
    if (tm * 256 + row < p.M) atomicAdd(acc_base + 2L * (tm * 256 + row) + which, (unsigned long long)sum);
}

// Residual tile streamed in during the k-loop (gemm256x_kernel<bf16, TC_EPI_BIAS_RESIDUAL, FOLD>).
// A 16-byte piece per lane comes straight into registers through a BOUNDED raw buffer descriptor (rows
// past the end of the plane read as zero, so a tile's row tail needs no clamping) by an assembly load:
// the compiler neither counts it nor waits for it -- the k-tile's one counted vmcnt wait covers it, because
// the loads are issued before that k-tile's first staging piece and the counter retires in issue order.
__device__ __forceinline__ u32x4 tc_raw_rsrc(const void *base, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    // (readfirstlane: the words must be provably wave-uniform to sit in the scalar registers the load takes them from)
    return u32x4{(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a),
                 (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu)),
                 (unsigned)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000u};
}
__device__ __forceinline__ u32x4 tc_buf_load16_async(u32x4 rsrc, unsigned lane_off, unsigned uniform_off) {
    u32x4 v;
    const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((int)uniform_off);
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v) : "v"(lane_off), "s"(rsrc), "s"(soff));
    return v;
}
// The same load into a NAMED register quadruple, and the matrix instruction that later reads it by name.  A value
// that is still on its way from memory when the compiler sees it "defined" must never be moved: with the
// register fixed at both ends there is no copy to insert.  Slot s = v[128 + 4 s .. 131 + 4 s].
template <int SLOT>
__device__ __forceinline__ void tc_buf_load16_slot(u32x4 &v, u32x4 rsrc, unsigned lane_off, unsigned uniform_off) {
    const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((int)uniform_off);
#define TC_SLOT_LOAD(S, R)                                                                                              \
    if constexpr (SLOT == S)                                                                                            \
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "={" R "}"(v) : "v"(lane_off), "s"(rsrc), "s"(soff))
    TC_SLOT_LOAD(0, "v[128:131]"); TC_SLOT_LOAD(1, "v[132:135]"); TC_SLOT_LOAD(2, "v[136:139]"); TC_SLOT_LOAD(3, "v[140:143]");
    TC_SLOT_LOAD(4, "v[144:147]"); TC_SLOT_LOAD(5, "v[148:151]"); TC_SLOT_LOAD(6, "v[152:155]"); TC_SLOT_LOAD(7, "v[156:159]");
#undef TC_SLOT_LOAD
}
template <int SLOT>
__device__ __forceinline__ void tc_mfma_bf16_slot(f32x4 &d, const f32x4 &sel, const u32x4 &v) {
#define TC_SLOT_MFMA(S, R)                                                                                              \
    if constexpr (SLOT == S) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(sel), "{" R "}"(v))
    TC_SLOT_MFMA(0, "v[128:131]"); TC_SLOT_MFMA(1, "v[132:135]"); TC_SLOT_MFMA(2, "v[136:139]"); TC_SLOT_MFMA(3, "v[140:143]");
    TC_SLOT_MFMA(4, "v[144:147]"); TC_SLOT_MFMA(5, "v[148:151]"); TC_SLOT_MFMA(6, "v[152:155]"); TC_SLOT_MFMA(7, "v[156:159]");
#undef TC_SLOT_MFMA
}
#ifndef TC_RES_STREAM_DEFAULT
#define TC_RES_STREAM_DEFAULT 1
#endif
// A tile's residual is 32 plane pieces (16 pieces of 16 rows x 32 columns per lane group, hi and lo plane each);
// STREAM of them are requested per k-tile during the tile's first ceil(32 / STREAM) k-tiles.
constexpr int res_stream_tiles(int per_ktile) { return (32 + per_ktile - 1) / per_ktile; }

constexpr bool piecewise_epilogue(int epi) {
    return epi == TC_EPI_BIAS || epi == TC_EPI_BIAS_GELU || epi == TC_EPI_BIAS_GELU_SAVE ||
           epi == TC_EPI_PLAIN_F32 || epi == TC_EPI_QKV_ROPE;
}

#ifdef TC_ABL_NOBAR
#define TC_XBAR() __builtin_amdgcn_sched_barrier(0)
#else
#define TC_XBAR()                                  \
    do {                                           \
        __builtin_amdgcn_sched_barrier(0);         \
        __builtin_amdgcn_s_barrier();              \
        __builtin_amdgcn_sched_barrier(0);         \
    } while (0)
#endif


template <typename T, int EPI, bool FOLD = false, bool PANELS = false, int STREAM = 0>
__global__ __launch_bounds__(512) void gemm256x_kernel(const GemmParams p) {
    static_assert(!STREAM || (FOLD && EPI == TC_EPI_BIAS_RESIDUAL && !PANELS), "streamed residual: the product on bf16 planes only");
    static_assert(STREAM == 0 || STREAM == 2 || STREAM == 3, "plane pieces per k-tile");
    using namespace k256x;
    static_assert(!PANELS || (EPI == TC_EPI_QKV_ROPE && FOLD), "row panel lists: folded q/k/v product only");
    static_assert(sizeof(T) == 2, "bf16 only");
    // Epilogues that read nothing from global memory (bias comes through LDS) run inside phase 0
    // of the next tile's first k-tile, after that phase's LDS-DMA issue and before its LDS reads:
    // nothing then waits for their stores.  The others run as one block between two tiles.
    // The residual product on bf16 planes has a deferred epilogue as well: its residual tile does not start
    // the accumulators but is added to them during the tile's first k-tiles, STREAM plane pieces per k-tile.
    constexpr bool kStream = STREAM > 0;
    constexpr bool kXchg = STREAM >= 3;          // residual requests coalesced by rows, lane exchange before use
    constexpr bool kPiece = piecewise_epilogue(EPI) || kStream;
    // bf16 results use the interleaved column order (16-byte stores); so does the f32 residual result
    // (two 16-byte stores side by side: 128 contiguous bytes per row, and its optional bf16 copy is one
    // 16-byte store); the plain f32 result keeps the plain order
    constexpr int kLay = EPI == TC_EPI_PLAIN_F32 ? 1 : 2;
    constexpr int kTileStores = (EPI == TC_EPI_BIAS_GELU_SAVE || kStream) ? 32 : (kLay == 2 ? 16 : 32);   // store instructions per wave
    constexpr bool kGeluTable = EPI == TC_EPI_BIAS_GELU || EPI == TC_EPI_BIAS_GELU_SAVE;
    static_assert(kPhiEntries * 4 <= STEP_OFF - ROPE_OFF, "the Phi table takes the place of the rotary slices");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wid >> 2, wc = wid & 3;                 // row group (= alternation group), column
    const int frow = lane & 15, fgrp = lane >> 4;

    // tile ownership: as in gemm256_kernel
    const int ntiles = p.tiles_m * p.tiles_n;
    const int cls = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int per_class = (gridDim.x + 7 - cls) >> 3;
    const int q = ntiles >> 3, r = ntiles & 7;
    const int first = cls < r ? cls * (q + 1) : r * (q + 1) + (cls - r) * q;
    const int count = q + (cls < r ? 1 : 0);
    const int my_count = count > j ? (count - j + per_class - 1) / per_class : 0;
    if (my_count == 0) return;
    // Tile order: column panel fastest inside a column group, all row panels of the first group (cg_w0 column
    // panels), then those of the second (the rest): with half the columns the weight panels an XCD works on
    // at one time (cg_w0 x 256 x K) stay in its L2 while the row panels stream by, instead of the whole
    // weight matrix cycling through it once per 32 tiles.  cg_w0 = 0: one group of all columns.
    // A workgroup's tiles are per_class apart, so (row panel, column panel) advances by a fixed step with one
    // carry: no division after this point except where a cursor crosses into the second group (each division
    // is a 100-clock dependent chain, and six cursors a tile would pay it).
    const int w0 = p.cg_w0 > 0 && p.cg_w0 < p.tiles_n ? p.cg_w0 : p.tiles_n;
    const int w1 = p.tiles_n - w0, w1s = w1 > 0 ? w1 : 1;
    const int base1 = p.tiles_m * w0;                       // linear index of the second group's first tile
    const int sm0 = per_class / w0, sn0 = per_class - sm0 * w0;
    const int sm1 = per_class / w1s, sn1 = per_class - sm1 * w1s;
    // kRev: the row panels are walked from the last one down (tm is always the panel's real index).  A product
    // that starts where its producer has just stopped finds the tail of its input still in the Infinity Cache:
    // fc1 runs against the direction of the residual product before it, and leaves its own tail at the low rows,
    // where fc2 starts.
    constexpr bool kRev = TC_GEMM_REV && (EPI == TC_EPI_BIAS_GELU || EPI == TC_EPI_BIAS_GELU_SAVE) && !PANELS;
    constexpr int kDir = kRev ? -1 : 1;
    auto locate = [&](int idx, int &tm, int &tn) {
        if (idx < base1) { tm = idx / w0; tn = idx - tm * w0; }
        else { const int l = idx - base1; tm = l / w1s; tn = w0 + l - tm * w1s; }
        if constexpr (kRev) tm = p.tiles_m - 1 - tm;
    };
    const int tile0 = first + j;
    int tm0, tn0;
    locate(tile0, tm0, tn0);
    auto advance = [&](int &tm, int &tn, int &idx) {
        const int nidx = idx + per_class;
        if (idx < base1) {
            if (nidx >= base1) locate(nidx, tm, tn);
            else { tn += sn0; tm += kDir * sm0; if (tn >= w0) { tn -= w0; tm += kDir; } }
        } else {
            tn += sn1; tm += kDir * sm1;
            if (tn >= p.tiles_n) { tn -= w1s; tm += kDir; }
        }
        idx = nidx;
    };

    // ---- staging: a half-tile is 16 pieces of 8 rows x 128 B; this wave moves pieces 2 wid, 2 wid + 1
    // The 16-byte chunk c of row r sits at chunk position c ^ key(r) of its LDS row.  A rows:
    // key = r & 7.  W rows, interleaved column order: key = (r & 3) | ((r >> 1) & 4), which is
    // what keeps the 8 rows one fragment read touches per cycle ({0..3, 8..11} + const) apart.
    const int srow = lane >> 3;
    const int schunk = (lane & 7) ^ srow;
    unsigned a1_src[2];                          // cursor 1 (k-tile g+1): A1
    unsigned b0_src[2], a0_src[2], b1_src[2];    // cursor 2 (k-tile g+2): B0, A0, B1
    // (with a streamed residual the lane's row and chunk are made afresh from the lane counter where a cursor
    // moves to its next tile, once per tile, instead of occupying registers through the k-loop)
    auto fresh_lane = [&]() -> int {
        int L = lane;
        if constexpr (kStream) {
            L = __builtin_amdgcn_mbcnt_hi(-1, __builtin_amdgcn_mbcnt_lo(-1, 0));
            asm volatile("" : "+v"(L));
        }
        return L;
    };
    auto a_src_of = [&](int tm, int h, int i) -> unsigned {
        int row0 = tm * BM, lim = p.M;
        if constexpr (PANELS) { row0 = p.panels[4 * tm]; lim = p.panels[4 * tm + 1]; }
        const int L = fresh_lane(), srow_ = kStream ? L >> 3 : srow, schunk_ = kStream ? (L & 7) ^ srow_ : schunk;
        int gm = row0 + h * 128 + (wid * 2 + i) * 8 + srow_;
        gm = gm < lim ? gm : lim - 1;
        return (unsigned)gm * (unsigned)p.lda_bytes + schunk_ * 16;
    };
    auto b_src_of = [&](int tn, int h, int i) -> unsigned {
        // row (2 wid + i) * 8 + srow of the half-tile: bit 3 of the row is i
        const int L = fresh_lane(), srow_ = kStream ? L >> 3 : srow, schunk_ = kStream ? (L & 7) ^ srow_ : schunk;
        const int chunk = kLay == 2 ? (L & 7) ^ ((srow_ & 3) | (i << 2)) : schunk_;
        return (unsigned)(tn * BN + h * 128 + (wid * 2 + i) * 8 + srow_) * (unsigned)p.ldw_bytes + chunk * 16;
    };
    int c1m = tm0, c1n = tn0, c2m = tm0, c2n = tn0;      // tiles of the two staging cursors
    int c1i = tile0, c2i = tile0, ti = tile0;            // ... and their linear indices (and the math cursor's)
    auto set_c1 = [&]() { a1_src[0] = a_src_of(c1m, 1, 0); a1_src[1] = a_src_of(c1m, 1, 1); };
    auto set_c2 = [&]() {
        const int tm = c2m, tn = c2n;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            b0_src[i] = b_src_of(tn, 0, i);
            a0_src[i] = a_src_of(tm, 0, i);
            b1_src[i] = b_src_of(tn, 1, i);
        }
    };
#if TC_GEMM_BUFLOAD
    const __amdgpu_buffer_rsrc_t rsrc_a = tc_buffer_rsrc(p.A), rsrc_w = tc_buffer_rsrc(p.W);
#endif
    // (cache policy of the operand loads: TC_LOADA_AUX / TC_LOADW_AUX, 2 = nt, streaming lines)
    using IsA = std::integral_constant<int, 1>;
    using IsW = std::integral_constant<int, 0>;
    auto stage_half = [&](auto which, int slot_off, const char *base, const unsigned (&src)[2], unsigned koff) {
#ifndef TC_ABL_NODMA
        char *dst = smem + slot_off + wid * 2048;
#if TC_GEMM_BUFLOAD
        constexpr bool kIsA = decltype(which)::value == 1;
        const __amdgpu_buffer_rsrc_t rsrc = kIsA ? rsrc_a : rsrc_w;
        constexpr int kAux = kIsA ? TC_LOADA_AUX : TC_LOADW_AUX;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, TC_LDS_PTR(dst), 16, (int)src[0], (int)koff, 0, kAux);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, TC_LDS_PTR(dst + 1024), 16, (int)src[1], (int)koff, 0, kAux);
#else
        glds16(base + (src[0] + koff), dst);
        glds16(base + (src[1] + koff), dst + 1024);
#endif
#endif
    };
    // This wave's private copies of what a tile's epilogue reads: staged in the tile's second
    // k-tile (before that k-tile's A1, so the phase-3 wait covers them), read after its last.
    const bool lds_bias = kPiece && p.bias != nullptr;
    char *const bias_slot = smem + BIAS_OFF + wid * 1024;
    char *const rope_slot = smem + ROPE_OFF + wid * 2048;
    char *const step_slot = smem + STEP_OFF + wid * 512;
    constexpr bool ln_fold = FOLD && EPI != TC_EPI_BIAS_RESIDUAL;   // a LayerNorm folded into this product (p.ln_rows, p.ln_s)
    static_assert(!FOLD || EPI == TC_EPI_QKV_ROPE || EPI == TC_EPI_BIAS_GELU || EPI == TC_EPI_BIAS_RESIDUAL,
                  "no folded form of this epilogue");
    // FOLD on the residual epilogue: the producer side in split form (residual stream as bf16 planes)
    constexpr bool kSplit = FOLD && EPI == TC_EPI_BIAS_RESIDUAL;
    char *const rows_slot = smem + ROWS_OFF + wr * 1024;
    auto stage_epilogue_inputs = [&](int tm, int tn) {
        int L = lane;
        asm volatile("" : "+v"(L));                // the address arithmetic below stays here, out of the k-loop's live set
        // lanes 0..7: the wave's 32 columns of the tile's first column half, 8..15: of the second;
        // lanes 16..31 the same columns of the folded weight's row sums
        if (lds_bias && L < (ln_fold ? 32 : 16)) {
            const float *vec = L < 16 ? p.bias : p.ln_s;
            glds16(reinterpret_cast<const char *>(vec + tn * BN + ((L >> 3) & 1) * 128 + wc * 32 + (L & 7) * 4), bias_slot);
        }
        if (ln_fold && wc == 0) {
            if constexpr (PANELS) {
                // a panel may start at an odd row: four pieces of 32 rows, one float of a (rstd, -rstd mean) pair
                // per lane (a 16-byte piece would need an even first row); reads stay inside the M rows
                const int row0 = p.panels[4 * tm];
                const long last = 2L * p.M - 1;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const long f = 2L * (row0 + (j >> 1) * 128 + wr * 64 + (j & 1) * 32) + L;
                    __builtin_amdgcn_global_load_lds(TC_GLOBAL_PTR(reinterpret_cast<const char *>(p.ln_rows + (f < last ? f : last))),
                                                     TC_LDS_PTR(rows_slot + j * 256), 4, 0, 0);
                }
            } else {
                // rows wr*64 .. +63 (lanes 0..31, two rows each) and 128 + wr*64 .. (lanes 32..63)
                const int r0 = tm * BM + (L >> 5) * 128 + wr * 64 + (L & 31) * 2;
                glds16(reinterpret_cast<const char *>(p.ln_rows + 2L * r0), rows_slot);
            }
        }
        if constexpr (EPI == TC_EPI_QKV_ROPE) {
            if ((tn * BN) / p.E < 2) {
                // lane (fgrp, frow): row tile_pos0 + wr*64 + frow, pairs ((wc*32 + fgrp*8) % 64) / 2 .. + 3
                int trow = (tm * BM) % p.T + wr * 64 + frow;
                if constexpr (PANELS) {
                    // positions past the end of a short last panel belong to rows that are not stored
                    trow = p.panels[4 * tm + 2] + wr * 64 + frow;
                    trow = trow < p.T ? trow : p.T - 1;
                }
                const long e = (long)trow * 32 + (((wc * 32 + fgrp * 8) & 63) >> 1);
                glds16(reinterpret_cast<const char *>(p.rope_cos + e), rope_slot);
                glds16(reinterpret_cast<const char *>(p.rope_sin + e), rope_slot + 1024);
            }
        }
    };

    // ---- streamed residual: plane descriptors, the lane's offset inside a tile, selection fragments ----
    // Piece (nq, mi) of a tile is 16 rows x 32 columns of a plane: lane (frow, fgrp) holds row frow, columns
    // 8 fgrp .. + 7, i.e. the B operand of mfma_f32_16x16x32_bf16 with k = column inside the 32-column block.
    // Multiplied by a 0/1 selection fragment S_nb (S_nb[i][k] = 1 where column k is operand row i of
    // accumulator tile nb: k = (i >> 2) * 8 + nb * 4 + (i & 3)) it adds the plane's values to the accumulator
    // pair of that block exactly (1.0 x bf16 products, f32 accumulation): the matrix unit does the residual
    // add, no conversion or vector add exists, and the loaded registers are MFMA operands as they stand.
    u32x4 rsrc_hi = {0u, 0u, 0u, 0u}, rsrc_lo = {0u, 0u, 0u, 0u};
    unsigned res_voff = 0;
    char *const ident_slot = smem + ROWS_OFF;     // S_0, S_1: 1 KiB each (the row-statistics slot is free here)
    if constexpr (kStream) {
        const unsigned plane_bytes = (unsigned)p.M * (unsigned)p.ldo * 2u;
        rsrc_hi = tc_raw_rsrc(p.residual, plane_bytes);
        rsrc_lo = tc_raw_rsrc(p.res_lo, plane_bytes);
        // coalesced: four neighbouring lanes read the 64 contiguous bytes of one row (lane L: row L >> 2, chunk L & 3);
        // a lane exchange after the wait brings them into the operand order (row frow, chunk fgrp).  Requests whose
        // sixteen neighbouring lanes go to sixteen different rows cost the memory pipeline about four times as much.
        // (kept for the short k-loops only, STREAM >= 3, where the memory pipeline is the bound: -2 % there; the long
        // ones sit on the LDS port, which the exchange uses too, and read in operand order: lane = frow + 16 fgrp)
        res_voff = kXchg ? (unsigned)((wr * 64 + (lane >> 2)) * p.ldo + wc * 32 + (lane & 3) * 8) * 2u
                         : (unsigned)((wr * 64 + frow) * p.ldo + wc * 32 + fgrp * 8) * 2u;
        if (wid == 0) {
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                u32x4 w = {0u, 0u, 0u, 0u};
                const int kk = nb * 4 + (frow & 3);
                const unsigned one = (kk & 1) ? 0x3F800000u : 0x00003F80u;
                if (fgrp == (frow >> 2)) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[e] = e == (kk >> 1) ? one : 0u;
                }
                *reinterpret_cast<u32x4 *>(ident_slot + nb * 1024 + lane * 16) = w;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }

    int a_off[2], b_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int c = ((ks * 4 + fgrp) ^ (frow & 7)) << 4;
        a_off[ks] = (wr * 64 + frow) * 128 + c;     // + i * 2048 inside half-tile A{mq}
        if constexpr (kLay == 2) {                  // + nb * 512: operand row frow of tile nb
            const int brow = (frow >> 2) * 8 + (frow & 3);
            const int key = (frow & 3) | (((frow >> 2) & 1) << 2);
            b_off[ks] = (wc * 32 + brow) * 128 + (((ks * 4 + fgrp) ^ key) << 4);
        } else {
            b_off[ks] = (wc * 32 + frow) * 128 + c; // + nb * 2048 inside half-tile B{nq}
        }
    }

    // ---- prologue: all of k-tile 0, and B0 A0 B1 of k-tile 1 (the host guarantees nk >= 2) ----
    set_c1();
    set_c2();
    if constexpr (EPI == TC_EPI_QKV_ROPE) {
        // rows 16 and 80 of the tables: 4 x 128 bytes, lanes 0..31
        if (lane < 32) {
            const float *tab = (lane & 8) ? p.rope_sin : p.rope_cos;
            glds16(reinterpret_cast<const char *>(tab + ((lane & 16) ? 80 : 16) * 32 + (lane & 7) * 4), step_slot);
        }
    }
    stage_half(IsW{}, SLOT_B0, p.W, b0_src, 0);
    stage_half(IsA{}, SLOT_A0, p.A, a0_src, 0);
    stage_half(IsW{}, SLOT_B1, p.W, b1_src, 0);
    stage_half(IsA{}, SLOT_A1, p.A, a1_src, 0);
    stage_half(IsW{}, DBUF_BYTES + SLOT_B0, p.W, b0_src, 128);
    stage_half(IsA{}, DBUF_BYTES + SLOT_A0, p.A, a0_src, 128);
    stage_half(IsW{}, DBUF_BYTES + SLOT_B1, p.W, b1_src, 128);
    int c1_tile = 0, c1_k = 1;                   // next A1 to stage
    int c2_tile = 0, c2_k = 2;                   // next B0 A0 B1 to stage
    if (c2_k == p.nk) { c2_k = 0; c2_tile = 1; if (c2_tile < my_count) { advance(c2m, c2n, c2i); set_c2(); } }
    if constexpr (kGeluTable) {
#pragma unroll
        for (int e = threadIdx.x; e < kPhiEntries; e += 512)
            *reinterpret_cast<float *>(smem + ROPE_OFF + e * 4) = phi_entry(e);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    TC_XBAR();
    // Row group 1 runs half a phase behind.  With a deferred epilogue the offset lasts for the whole
    // kernel; otherwise (epilogues that load from global memory and wait for it) the groups meet
    // again after each tile's k-loop, run their epilogues and accumulator loads side by side, and
    // part again at the next tile's first phase: the two memory waits overlap instead of adding up.
    if constexpr (kPiece) {
        if (wr == 1) TC_XBAR();
    }

    f32x4 acc[4][8];                              // [ni][mi], D[n][m]
    u32x4 rq[2][4] = {};                          // streamed residual: two register sets of plane pieces in flight
    if constexpr (kPiece) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // (The deferred epilogue as four quarters, each in the load slot of the phase whose MFMAs use that quarter's
    // accumulators next, was measured 4-11 % slower on the K = 768 shapes: a wave's epilogue is bound by the latency
    // of its exchange / store chains, which the two waves of a SIMD hide for each other when they run it side by
    // side and a quarter beside the partner's MFMAs cannot.)
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    // epilogue of tile `local` (bias from this wave's LDS copy), accumulators back to zero
    auto finish_tile = [&](int tm, int tn, int parity) __attribute__((always_inline)) {
        const int mrow0 = tm * BM + wr * 64 + frow, ncol0 = tn * BN + wc * 32 + fgrp * (kLay == 2 ? 8 : 4);
        if constexpr (kStream) {
            // bias from the wave's LDS slot, the result as bf16 hi / lo planes, row sums of the hi plane into
            // the plane set of this tile's parity (flushed one k-tile later, behind that k-tile's barriers)
            epilogue256<T, EPI, kLay, 0, 8, 0, 4, false, true>(
                p, acc, tm, tn, mrow0, ncol0, wr, wc, frow, fgrp, lds_bias ? bias_slot : nullptr, nullptr, nullptr, nullptr,
                p.row_acc ? smem + ROPE_OFF + parity * 8192 : nullptr);
        } else if constexpr (PANELS) {
            // the epilogue addresses row 256 tm + r of `out` and keeps rows below M: shift both to the panel
            GemmParams q = p;
            const int shift = p.panels[4 * tm] - tm * BM;
            q.out = p.out + (long)shift * p.ldo * 2;
            q.M = p.panels[4 * tm + 1] - shift;
            epilogue256<T, EPI, kLay, 0, 8, 0, 4, true>(q, acc, tm, tn, mrow0, ncol0, wr, wc, frow, fgrp, bias_slot, rope_slot,
                                                        step_slot, rows_slot);
        } else
        epilogue256<T, EPI, kLay, 0, 8, 0, 4, EPI == TC_EPI_QKV_ROPE || kGeluTable>(
            p, acc, tm, tn, mrow0, ncol0, wr, wc, frow, fgrp, lds_bias ? bias_slot : nullptr,
            kGeluTable ? smem + ROPE_OFF : rope_slot, step_slot, ln_fold ? rows_slot : nullptr);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
    };

#ifdef TC_GEMM_STAMPS
    unsigned long long c_kt[24] = {}, c_epi = 0;
    TC_STAMP(k_begin);
    const unsigned long long rt_begin = __builtin_amdgcn_s_memrealtime();   // 100 MHz: gives the clock the kernel ran at
#endif
#ifdef TC_GEMM_PHASES
    // Hand-over profile: only two k-tile positions of a tile carry stamps (13 points each; the first and the
    // last unless -DTC_PH_KA= / -DTC_PH_KB= name others); slots [A / B][interval] accumulate shader clocks
    // between consecutive points.
#ifndef TC_PH_KA
#define TC_PH_KA 0
#endif
#ifndef TC_PH_KB
#define TC_PH_KB (p.nk - 1)
#endif
    unsigned c_ph[2][13] = {};
    unsigned long long ph_t = 0;
#define TC_PH(i)                                                                              \
    do {                                                                                      \
        if (kt == TC_PH_KA || kt == TC_PH_KB) {                                               \
            const unsigned long long now__ = __builtin_amdgcn_s_memtime();                    \
            c_ph[kt == TC_PH_KA ? 0 : 1][i] += (unsigned)(now__ - ph_t);                      \
            ph_t = now__;                                                                     \
        }                                                                                     \
    } while (0)
#define TC_PH_START() do { if (kt == TC_PH_KA || kt == TC_PH_KB) ph_t = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TC_PH(i)
#define TC_PH_START()
#endif
    int g = 0;
    int tm = tm0, tn = tn0, ptm = tm0, ptn = tn0;          // this tile and the one before it
    for (int t = 0; t < my_count; ++t) {
        const int mrow0 = tm * BM + wr * 64 + frow;           // + mi_off<1>(mi)
        const int ncol0 = tn * BN + wc * 32 + fgrp * (kLay == 2 ? 8 : 4);     // + ni_off<kLay>(ni)
        if constexpr (!kPiece) {
            init_acc256<T, EPI, kLay, 0, 8, 0, 4, kSplit>(p, acc, mrow0, ncol0);
            if (wr == 1) TC_XBAR();
        }
        // every store instruction of the previous tile's epilogue was issued, row tail or not (bounded descriptors
        // drop the rows past the end), so the counted waits below always hold
        const bool prev_full = kPiece && t > 0;

        // The unrolled k-tiles of the streamed form name the accumulators in fixed registers at every k-tile
        // boundary: without that the register allocator renames them from copy to copy of the body and pays
        // for it with moves and spills inside the loop.
        auto pin_acc = [&]() __attribute__((always_inline)) {
            asm volatile("" : "+{v[0:3]}"(acc[0][0]), "+{v[4:7]}"(acc[0][1]), "+{v[8:11]}"(acc[0][2]), "+{v[12:15]}"(acc[0][3]), "+{v[16:19]}"(acc[0][4]), "+{v[20:23]}"(acc[0][5]), "+{v[24:27]}"(acc[0][6]), "+{v[28:31]}"(acc[0][7]));
            asm volatile("" : "+{v[32:35]}"(acc[1][0]), "+{v[36:39]}"(acc[1][1]), "+{v[40:43]}"(acc[1][2]), "+{v[44:47]}"(acc[1][3]), "+{v[48:51]}"(acc[1][4]), "+{v[52:55]}"(acc[1][5]), "+{v[56:59]}"(acc[1][6]), "+{v[60:63]}"(acc[1][7]));
            asm volatile("" : "+{v[64:67]}"(acc[2][0]), "+{v[68:71]}"(acc[2][1]), "+{v[72:75]}"(acc[2][2]), "+{v[76:79]}"(acc[2][3]), "+{v[80:83]}"(acc[2][4]), "+{v[84:87]}"(acc[2][5]), "+{v[88:91]}"(acc[2][6]), "+{v[92:95]}"(acc[2][7]));
            asm volatile("" : "+{v[96:99]}"(acc[3][0]), "+{v[100:103]}"(acc[3][1]), "+{v[104:107]}"(acc[3][2]), "+{v[108:111]}"(acc[3][3]), "+{v[112:115]}"(acc[3][4]), "+{v[116:119]}"(acc[3][5]), "+{v[120:123]}"(acc[3][6]), "+{v[124:127]}"(acc[3][7]));
        };
        // One k-tile.  J >= 0 (kStream): the J-th of a tile's first res_stream_tiles(STREAM) k-tiles, instantiated once
        // each so that the two residual pieces it adds name their accumulators statically; J = -1: any other.
        auto ktile = [&](const int kt, auto jc) __attribute__((always_inline)) {
            constexpr int J = decltype(jc)::value;
            if constexpr (kStream) pin_acc();
            const int cur = (g & 1) * DBUF_BYTES, nxt = DBUF_BYTES - cur;
            const char *sb = smem + cur;
            // the epilogue of tile t-1 is pending (with a streamed residual only instance J = 0 is ever a first k-tile)
            const bool after = kPiece && (!kStream || J == 0) && kt == 0 && t > 0;
            TC_STAMP(s0);
            f32x4 bq[2][2][2];                   // [nq][column block][k-step]
            f32x4 af[4][2];                      // [row block][k-step] of the current mq
            // Streamed residual.  Instance J < kU requests plane pieces STREAM J .. STREAM J + STREAM - 1 of this tile
            // (plane piece q: column half q >> 4, row block (q >> 1) & 7, plane q & 1; 16 bytes per lane) in its
            // phase 3, into register set J & 1; the counted wait of that phase leaves them in flight with the three
            // half-tiles issued after this k-tile's A1, and the wait of the NEXT k-tile (instance J + 1) covers
            // them: one k-tile of lead.  Instance J + 1 multiplies them into their accumulator pairs at the start
            // of its phase 3 MFMA turn.
            constexpr int JJ = J < 0 ? 0 : J;
            constexpr int kU = kStream ? res_stream_tiles(STREAM ? STREAM : 1) : 0;
            constexpr int kIssued = (kStream && J >= 0 && J < kU) ? (32 - JJ * STREAM < STREAM ? 32 - JJ * STREAM : STREAM) : 0;
            auto res_issue = [&](auto uc) __attribute__((always_inline)) {
                constexpr int u = decltype(uc)::value, q = JJ * STREAM + u;
                if constexpr (u < kIssued) {
                    constexpr int nq = q >> 4, mi = (q >> 1) & 7;
                    const unsigned off = ((unsigned)(tm * BM + mi_off<2>(mi)) * (unsigned)p.ldo + (unsigned)(tn * BN + nq * 128)) * 2u;
                    tc_buf_load16_slot<(JJ & 1) * 4 + u>(rq[JJ & 1][u], (q & 1) ? rsrc_lo : rsrc_hi, res_voff, off);
                }
            };

            // A phase's 16 MFMAs, then the barrier that hands the matrix pipe to the other wave of the SIMD.
            // (Entering that barrier a few MFMAs before the end of the turn removes the wake-up gap of the released
            // wave on bare MFMA turns, tools/ubench/mfma_handover.hip: 256 -> 217 ns per 32 MFMAs; here it cost 3-9 %,
            // because the phase is bound by the load slot of the wave that is not multiplying, and the late MFMAs
            // delay the start of that slot.)
            auto math = [&](int mq, int nq, bool hand_over = true) {
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            acc[nq * 2 + nb][mq * 4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                __builtin_bit_cast(bf16x8, bq[nq][nb][ks]), __builtin_bit_cast(bf16x8, af[i][ks]),
                                acc[nq * 2 + nb][mq * 4 + i], 0, 0, 0);
                        }
                __builtin_amdgcn_s_setprio(0);
                if (hand_over) TC_XBAR();
            };
            auto read_b = [&](int nq) {
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
#if defined(TC_ABL_NOREADS) || defined(TC_ABL_NOREADS_B)
                        bq[nq][nb][ks] = f32x4{(float)kt, 1.f, 2.f, 3.f};
                        asm volatile("" : "+v"(bq[nq][nb][ks]));
#else
                        bq[nq][nb][ks] = *reinterpret_cast<const f32x4 *>(
                            sb + (nq ? SLOT_B1 : SLOT_B0) + b_off[ks] + nb * (kLay == 2 ? 512 : 2048));
#endif
                    }
            };
            auto read_a = [&](int mq) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
#ifdef TC_ABL_NOREADS
                        af[i][ks] = f32x4{(float)kt, 1.f, 2.f, 3.f};
                        asm volatile("" : "+v"(af[i][ks]));
#else
                        af[i][ks] = *reinterpret_cast<const f32x4 *>(
                            sb + (mq ? SLOT_A1 : SLOT_A0) + a_off[ks] + i * 2048);
#endif
                    }
            };

            // Order inside a phase: LDS-DMA issue, then (first k-tile only) the previous tile's
            // epilogue, while no operand fragment is live, then the LDS reads.
            // ---- phase 0 ----
            TC_PH_START();
            if (kPiece && kt == 1) stage_epilogue_inputs(tm, tn);
            if constexpr (kStream) {
                // row sums of the tile before: its epilogue ran one k-tile ago, the barriers since then have
                // published every wave's plane entries (an atomic per thread, older than this k-tile's A1)
                if (kt == 1 && t > 0 && p.row_acc) flush_row_sums<true>(p, smem + ROPE_OFF + ((t - 1) & 1) * 8192, ptm);
            }
            if (c1_tile < my_count) {
                stage_half(IsA{}, nxt + SLOT_A1, p.A, a1_src, (unsigned)c1_k * 128u);
                if (++c1_k == p.nk) { c1_k = 0; if (++c1_tile < my_count) { advance(c1m, c1n, c1i); set_c1(); } }
            }
            if constexpr (kPiece) {
                if (after) {
                    TC_STAMP(e0);
                    finish_tile(ptm, ptn, (t - 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);
#ifdef TC_GEMM_STAMPS
                    TC_STAMP(e1);
                    c_epi += e1 - e0;
#endif
                    if (wr == 1) TC_XBAR();        // the two groups ran this epilogue side by side: part again
                }
            }
            TC_PH(0);                              // DMA issue + previous tile's epilogue
            read_b(0);
            __builtin_amdgcn_sched_barrier(0);
            read_a(0);
            asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
            TC_PH(1);                              // reads issued
            TC_XBAR();
            TC_PH(2);                              // wait at barrier 1
            math(0, 0);                            // ends with the hand-over barrier
            TC_PH(3);                              // MFMA + barrier 2
            // ---- phase 1 ----
            const bool more = c2_tile < my_count;
            const unsigned koff2 = (unsigned)c2_k * 128u;
            if (more) stage_half(IsW{}, cur + SLOT_B0, p.W, b0_src, koff2);
            read_b(1);
            TC_PH(4);
            TC_XBAR();
            TC_PH(5);
            math(0, 1);                            // ends with the hand-over barrier
            TC_PH(6);
            // ---- phase 2 ----
            if (more) stage_half(IsA{}, cur + SLOT_A0, p.A, a0_src, koff2);
            read_a(1);
            TC_PH(7);
            TC_XBAR();
            TC_PH(8);
            math(1, 1);                            // ends with the hand-over barrier
            TC_PH(9);
            // ---- phase 3 ----
            // The wait leaves in flight what was issued after this k-tile's A1: three half-tiles,
            // plus, in a tile's first k-tile, the previous tile's stores (when none of its rows
            // was masked, so that every store instruction was issued; the counter holds 63).
            // Streamed residual: this k-tile's requests and the selection fragments for the pieces that are about to
            // be added go out here, in the lightest load slot of the four (two staging pieces, no fragment reads)
            f32x4 s0f = {0.f, 0.f, 0.f, 0.f}, s1f = s0f;
            if constexpr (kIssued > 0) {
                res_issue(std::integral_constant<int, 0>{});
                res_issue(std::integral_constant<int, 1>{});
                res_issue(std::integral_constant<int, 2>{});
                res_issue(std::integral_constant<int, 3>{});
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (kStream && J >= 1) {
                const int L16 = fresh_lane() * 16;
                s0f = *reinterpret_cast<const f32x4 *>(ident_slot + L16);
                s1f = *reinterpret_cast<const f32x4 *>(ident_slot + 1024 + L16);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (more) {
                stage_half(IsW{}, cur + SLOT_B1, p.W, b1_src, koff2);
                if (++c2_k == p.nk) { c2_k = 0; if (++c2_tile < my_count) { advance(c2m, c2n, c2i); set_c2(); } }
                // (kIssued: this k-tile's residual plane pieces, younger than its A1 as well)
                if (after && prev_full) {
                    if constexpr (kTileStores == 32) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(38 + kIssued) : "memory");
                    else asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 + kIssued) : "memory");
                }
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            TC_PH(10);                             // DMA issue + vmcnt wait
            TC_XBAR();
            TC_PH(11);
            if constexpr (kStream && J >= 1) {
                // The plane pieces requested one k-tile ago have landed (the counted wait above).  Lane exchange into
                // the operand order first (lane frow + 16 fgrp takes what lane 4 frow + fgrp loaded), its latency
                // under this turn's sixteen MFMAs; then the adds, before the hand-over barrier.
                constexpr int JP = J - 1;
                const int Lx = fresh_lane();
                const int pull = (((Lx & 15) << 2) | (Lx >> 4)) << 2;
                auto res_xchg = [&](auto uc) __attribute__((always_inline)) {
                    constexpr int u = decltype(uc)::value, q = JP * STREAM + u;
                    if constexpr (kXchg && u < STREAM && q < 32) {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            rq[JP & 1][u][e] = (unsigned)__builtin_amdgcn_ds_bpermute(pull, (int)rq[JP & 1][u][e]);
                    }
                };
                res_xchg(std::integral_constant<int, 0>{}); res_xchg(std::integral_constant<int, 1>{});
                res_xchg(std::integral_constant<int, 2>{}); res_xchg(std::integral_constant<int, 3>{});
                __builtin_amdgcn_sched_barrier(0);
                math(1, 0, false);
                __builtin_amdgcn_sched_barrier(0);
                // in place, by name: D = C is not a constraint of the builtin, and the allocator otherwise moves
                // the accumulators through spare registers here.  hi planes first, then lo planes: the two adds
                // into one accumulator are not neighbours in the matrix pipe.
                auto res_add = [&](auto uc, auto plane_c) __attribute__((always_inline)) {
                    constexpr int u = decltype(uc)::value, q = JP * STREAM + u;
                    if constexpr (u < STREAM && q < 32 && (q & 1) == decltype(plane_c)::value) {
                        constexpr int nq = q >> 4, mi = (q >> 1) & 7;
                        // (these reads are what keeps the named destination registers of the loads above live:
                        // without a reader the allocator hands them to addresses while the loads are in flight)
                        tc_mfma_bf16_slot<(JP & 1) * 4 + u>(acc[nq * 2][mi], s0f, rq[JP & 1][u]);
                        tc_mfma_bf16_slot<(JP & 1) * 4 + u>(acc[nq * 2 + 1][mi], s1f, rq[JP & 1][u]);
                    }
                };
                res_add(std::integral_constant<int, 0>{}, I0{}); res_add(std::integral_constant<int, 1>{}, I0{});
                res_add(std::integral_constant<int, 2>{}, I0{}); res_add(std::integral_constant<int, 3>{}, I0{});
                res_add(std::integral_constant<int, 0>{}, I1{}); res_add(std::integral_constant<int, 1>{}, I1{});
                res_add(std::integral_constant<int, 2>{}, I1{}); res_add(std::integral_constant<int, 3>{}, I1{});
                __builtin_amdgcn_sched_barrier(0);
                TC_XBAR();
            } else {
                math(1, 0);                        // ends with the hand-over barrier
            }
            TC_PH(12);
#ifdef TC_GEMM_STAMPS
            TC_STAMP(s1);
#pragma unroll
            for (int i = 0; i < 24; ++i) if (i == (kt < 23 ? kt : 23)) c_kt[i] += s1 - s0;
#endif
            ++g;
        };
        using Plain = std::integral_constant<int, -1>;
        if constexpr (kStream) {
            // kU requesting instances and one more that only adds: the host guarantees nk > res_stream_tiles(STREAM)
            constexpr int kU = res_stream_tiles(STREAM);
            ktile(0, std::integral_constant<int, 0>{});
            ktile(1, std::integral_constant<int, 1>{});
            ktile(2, std::integral_constant<int, 2>{});
            ktile(3, std::integral_constant<int, 3>{});
            ktile(4, std::integral_constant<int, 4>{});
            ktile(5, std::integral_constant<int, 5>{});
            ktile(6, std::integral_constant<int, 6>{});
            ktile(7, std::integral_constant<int, 7>{});
            ktile(8, std::integral_constant<int, 8>{});
            if constexpr (kU > 8) {
                ktile(9, std::integral_constant<int, 9>{});
                ktile(10, std::integral_constant<int, 10>{});
                ktile(11, std::integral_constant<int, 11>{});
            }
            if constexpr (kU > 11) {
                ktile(12, std::integral_constant<int, 12>{});
                ktile(13, std::integral_constant<int, 13>{});
                ktile(14, std::integral_constant<int, 14>{});
                ktile(15, std::integral_constant<int, 15>{});
                ktile(16, std::integral_constant<int, 16>{});
            }
            for (int kt = kU + 1; kt < p.nk; ++kt) ktile(kt, Plain{});
        } else {
            for (int kt = 0; kt < p.nk; ++kt) ktile(kt, Plain{});
        }
        if constexpr (!kPiece) {
            if (wr == 0) TC_XBAR();               // row group 1's last phase
            TC_STAMP(e0);
            // row sums (folded LayerNorm, producer side): the planes of tile t-1 are flushed here, one
            // tile late, so that no barrier is needed between the waves' plane writes and the flush
            // (the k-loop's barriers lie in between); two sets of planes alternate
            char *const stat_planes = smem + BIAS_OFF + (t & 1) * 8192;
            if constexpr (EPI == TC_EPI_BIAS_RESIDUAL) {
                if (p.row_acc && t > 0) flush_row_sums(p, smem + BIAS_OFF + ((t - 1) & 1) * 8192, ptm);
            }
            epilogue256<T, EPI, kLay, 0, 8, 0, 4, kGeluTable, kSplit>(p, acc, tm, tn, mrow0, ncol0, wr, wc, frow, fgrp, nullptr,
                                                              smem + ROPE_OFF, nullptr, nullptr,
                                                              p.row_acc ? stat_planes : nullptr);
#ifdef TC_GEMM_STAMPS
            TC_STAMP(e1);
            c_epi += e1 - e0;
#endif
        }
        // After a tile's last phase row group 0 waits for row group 1, so that both run the
        // deferred epilogue at the same time (each hides the other's LDS and exchange latency)
        // instead of one after the other; group 1 falls half a phase behind again right after it.
        if constexpr (kPiece) {
            if (wr == 0) TC_XBAR();
        }
        ptm = tm;
        ptn = tn;
        advance(tm, tn, ti);
    }
#ifdef TC_GEMM_STAMPS
    if (p.stamps && (threadIdx.x == 0 || threadIdx.x == 256)) {
        TC_STAMP(k_end);
        unsigned long long *o = p.stamps + (blockIdx.x * 2 + (threadIdx.x >> 8)) * 32;
        for (int i = 0; i < 24; ++i) o[i] = c_kt[i];
        o[24] = c_epi; o[25] = k_end - k_begin; o[26] = my_count;
        o[27] = __builtin_amdgcn_s_memrealtime() - rt_begin;
#ifdef TC_GEMM_PHASES
        unsigned long long *o2 = p.stamps + 256 * 2 * 32 + (blockIdx.x * 2 + (threadIdx.x >> 8)) * 26;
        for (int c = 0; c < 2; ++c)
            for (int i = 0; i < 13; ++i) o2[c * 13 + i] = c_ph[c][i];
#endif
    }
#endif
    if constexpr (kPiece) finish_tile(ptm, ptn, (my_count - 1) & 1);
    if constexpr (EPI == TC_EPI_BIAS_RESIDUAL) {
        if (p.row_acc) {                          // the last tile's row sums
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_s_barrier();
            flush_row_sums(p, smem + (kStream ? ROPE_OFF : BIAS_OFF) + ((my_count - 1) & 1) * 8192, ptm);
        }
    }
}

// =============================================================================================
// host side
// =============================================================================================
int g_num_cus = 0;
#ifdef TC_GEMM_STAMPS
unsigned long long *g_stamps = nullptr;
#endif

// TC_GEMM_ALT=0 selects the single-group schedule of the large tile
int alt_schedule_enabled() {
    static int alt = -1;
    if (alt < 0) { const char *e = getenv("TC_GEMM_ALT"); alt = e ? atoi(e) != 0 : 1; }
    return alt;
}

template <typename T, int EPI>
int launch(const GemmParams &p0, bool big, hipStream_t stream) {
    GemmParams p = p0;
    if (big) {
        if constexpr (sizeof(T) == 2) {
        static bool attr_set = false;   // benign race: the attribute calls are idempotent
        auto kern = gemm256_kernel<T, EPI>;
        const int lds = k256::LDS_BYTES;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(gemm256_kernel<T, EPI>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, k256::LDS_BYTES);
            if (e != hipSuccess) {
                tc_set_error("tc_gemm: hipFuncSetAttribute: %s", hipGetErrorString(e));
                return TC_ELAUNCH;
            }
            attr_set = true;
        }
        p.tiles_m = p.panels ? p.n_panels : (p.M + k256::BM - 1) / k256::BM;
        p.tiles_n = p.N / k256::BN;
        const int ntiles = p.tiles_m * p.tiles_n;
        int grid = ntiles < g_num_cus ? ntiles : g_num_cus;
        {
            static int cap = -1;
            if (cap < 0) { const char *e = getenv("TC_GEMM_GRID"); cap = e ? atoi(e) : 0; }
            if (cap > 0 && cap < grid) grid = cap;
        }
        {
            static int gm = 0;
            if (gm == 0) { const char *e = getenv("TC_GEMM_GM"); gm = e ? atoi(e) : 1; if (gm < 1) gm = 1; }
            p.gm = gm;
        }
        {
            // Optional column grouping (TC_GEMM_GN=g): lowers the L2 fill traffic of the wide-N
            // shapes (the W panels of a group stay in the XCD's L2) but measured 1 % slower in the
            // whole forward, because the re-reads it saves are Infinity Cache hits that the
            // kernel's load path hides anyway.  Off by default.
            static int force = -2;
            if (force == -2) { const char *e = getenv("TC_GEMM_GN"); force = e ? atoi(e) : -1; }
            p.gn = force > 0 && force < p.tiles_n ? force : p.tiles_n;
            if (p.gm > 1) p.gn = p.tiles_n;
        }
        {
            // alternating kernel: two column groups from 10 column panels on (fc1: its 4.7 MB of weights do not fit an
            // XCD's L2; q/k/v with 9 panels and 3.5 MB ran 5-7 % slower grouped); TC_GEMM_CG=0 turns it off
            static int cg = -2;
            if (cg == -2) { const char *e = getenv("TC_GEMM_CG"); cg = e ? atoi(e) : -1; }
            p.cg_w0 = cg == 0 ? 0 : (cg > 0 ? cg : (p.tiles_n >= 10 ? (p.tiles_n + 1) / 2 : 0));
        }
#ifdef TC_GEMM_STAMPS
        p.stamps = g_stamps;
        { const char *e = getenv("TC_GEMM_NOSTORE"); p.nostore = e ? atoi(e) : 0; }
#endif
        const int alt = alt_schedule_enabled();
        // the alternating kernel's rotary epilogue steps through a tile by angle addition: no
        // sequence boundary inside a 256-row tile, head dimension 64
        bool use_alt = alt != 0;
        if (EPI == TC_EPI_QKV_ROPE && ((p.T % 256 != 0 && !p.panels) || p.head_dim != 64 || p.pos)) use_alt = false;
        if (p.panels) {
            // gemm_impl has checked: folded q/k/v product on the alternating schedule
            if constexpr (EPI == TC_EPI_QKV_ROPE) {
                static bool attr_set_p = false;
                if (!attr_set_p) {
                    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(gemm256x_kernel<T, EPI, true, true>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, k256x::LDS_BYTES);
                    if (e != hipSuccess) {
                        tc_set_error("tc_gemm: hipFuncSetAttribute: %s", hipGetErrorString(e));
                        return TC_ELAUNCH;
                    }
                    attr_set_p = true;
                }
                TC_LAUNCH((gemm256x_kernel<T, EPI, true, true>), dim3(grid), dim3(512), k256x::LDS_BYTES, stream, p);
            }
            TC_CHECK_LAUNCH("tc_gemm");
            return TC_OK;
        }
        constexpr bool kHasFold = EPI == TC_EPI_QKV_ROPE || EPI == TC_EPI_BIAS_GELU || EPI == TC_EPI_BIAS_RESIDUAL;
        if (use_alt && kHasFold && (EPI == TC_EPI_BIAS_RESIDUAL ? p.res_lo != nullptr : p.ln_rows != nullptr)) {
            if constexpr (kHasFold) {
                static bool attr_set_f = false;
                if (!attr_set_f) {
                    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(gemm256x_kernel<T, EPI, true>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, k256x::LDS_BYTES);
                    if (e != hipSuccess) {
                        tc_set_error("tc_gemm: hipFuncSetAttribute: %s", hipGetErrorString(e));
                        return TC_ELAUNCH;
                    }
                    attr_set_f = true;
                }
                // the residual planes streamed in under the k-loop: eight k-tiles or more to carry the sixteen pieces of
                // a tile, plane offsets in 32 bits; TC_RES_STREAM=0 keeps the form that preloads the accumulators
                bool streamed = false;
                if constexpr (EPI == TC_EPI_BIAS_RESIDUAL) {
                    static int want = -1;
                    if (want < 0) { const char *e = getenv("TC_RES_STREAM"); want = e ? atoi(e) : TC_RES_STREAM_DEFAULT; }
                    // plane pieces per k-tile: as few as the k-tile count allows (want > 1 forces a count)
                    // three plane pieces per k-tile whatever K is (in the forward's kernel trace fc2 runs 0.5 % faster
                    // with three pieces over eleven k-tiles than with two over sixteen); TC_RES_STREAM=2 forces two
                    int per = 3;
                    if (want >= 2 && want <= 3 && p.nk > res_stream_tiles(want)) per = want;
                    streamed = want != 0 && p.nk > res_stream_tiles(per) && (long)p.M * p.ldo * 2 < (1L << 32);
                    if (streamed) {
                        auto go = [&](auto kern, bool &attr_set) -> int {
                            if (!attr_set) {
                                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                                                   hipFuncAttributeMaxDynamicSharedMemorySize, k256x::LDS_BYTES);
                                if (e != hipSuccess) {
                                    tc_set_error("tc_gemm: hipFuncSetAttribute: %s", hipGetErrorString(e));
                                    return TC_ELAUNCH;
                                }
                                attr_set = true;
                            }
                            TC_LAUNCH(kern, dim3(grid), dim3(512), k256x::LDS_BYTES, stream, p);
                            return TC_OK;
                        };
                        static bool set2 = false, set3 = false;
                        int rc = TC_OK;
                        if (per == 2) rc = go(gemm256x_kernel<T, EPI, true, false, 2>, set2);
                        else rc = go(gemm256x_kernel<T, EPI, true, false, 3>, set3);
                        if (rc != TC_OK) return rc;
                    }
                }
                if (!streamed) TC_LAUNCH((gemm256x_kernel<T, EPI, true>), dim3(grid), dim3(512), k256x::LDS_BYTES, stream, p);
            }
        } else if (use_alt && tc_gemm256w_supported(EPI, p)) {
            return tc_gemm256w_launch(EPI, p, grid, stream);
        } else if (use_alt) {
            static bool attr_set_x = false;
            if (!attr_set_x) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(gemm256x_kernel<T, EPI>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, k256x::LDS_BYTES);
                if (e != hipSuccess) {
                    tc_set_error("tc_gemm: hipFuncSetAttribute: %s", hipGetErrorString(e));
                    return TC_ELAUNCH;
                }
                attr_set_x = true;
            }
            TC_LAUNCH((gemm256x_kernel<T, EPI>), dim3(grid), dim3(512), k256x::LDS_BYTES, stream, p);
        } else {
            TC_LAUNCH(kern, dim3(grid), dim3(512), lds, stream, p);
        }
        }
    } else {
        static bool attr_set = false;
        auto kern = gemm128_kernel<T, EPI>;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize,
                                               k128::LDS_BYTES);
            if (e != hipSuccess) {
                tc_set_error("tc_gemm: hipFuncSetAttribute: %s", hipGetErrorString(e));
                return TC_ELAUNCH;
            }
            attr_set = true;
        }
        p.tiles_m = (p.M + k128::BM - 1) / k128::BM;
        p.tiles_n = p.N / k128::BN;
        TC_LAUNCH(kern, dim3(p.tiles_m * p.tiles_n, p.splits), dim3(256), k128::LDS_BYTES, stream, p);
    }
    TC_CHECK_LAUNCH("tc_gemm");
    return TC_OK;
}

template <typename T>
int dispatch(int epilogue, const GemmParams &p, bool big, hipStream_t stream) {
    switch (epilogue) {
        case TC_EPI_BIAS: return launch<T, TC_EPI_BIAS>(p, big, stream);
        case TC_EPI_BIAS_GELU: return launch<T, TC_EPI_BIAS_GELU>(p, big, stream);
        case TC_EPI_BIAS_RESIDUAL: return launch<T, TC_EPI_BIAS_RESIDUAL>(p, big, stream);
        case TC_EPI_QKV_ROPE: return launch<T, TC_EPI_QKV_ROPE>(p, big, stream);
        case TC_EPI_PLAIN_F32: return launch<T, TC_EPI_PLAIN_F32>(p, big, stream);
        case TC_EPI_BIAS_GELU_SAVE: return launch<T, TC_EPI_BIAS_GELU_SAVE>(p, big, stream);
        case TC_EPI_DGELU: return launch<T, TC_EPI_DGELU>(p, big, stream);
    }
    tc_set_error("tc_gemm: unknown epilogue %d", epilogue);
    return TC_EINVAL;
}

}  // namespace

#ifdef TC_GEMM_STAMPS
extern "C" void tc_debug_set_stamp_buffer(void *p) { g_stamps = static_cast<unsigned long long *>(p); }
#endif

static int gemm_impl(int dtype, int epilogue, const void *A, int lda, const void *W, const float *bias,
                     const float *residual, const void *aux, void *out, void *out2, int ldo, int M, int N,
                     int K, int splits, float q_scale, const float *rope_cos, const float *rope_sin, int T,
                     int head_dim, void *stream, const int32_t *pos = nullptr, const float *ln_s = nullptr,
                     const float *ln_rows = nullptr, unsigned long long *row_acc = nullptr, void *split_lo = nullptr,
                     const int32_t *panels = nullptr, int n_panels = 0) {
    TC_REQUIRE(dtype == TC_BF16 || dtype == TC_F32, "tc_gemm: bad dtype %d", dtype);
    const int esz = dtype == TC_BF16 ? 2 : 4;
    const int kt = 128 / esz;
    TC_REQUIRE(A && W && out, "tc_gemm: null operand");
    TC_REQUIRE(M > 0 && N > 0 && K > 0, "tc_gemm: empty problem M=%d N=%d K=%d", M, N, K);
    TC_REQUIRE(N % 128 == 0, "tc_gemm: N=%d must be a multiple of 128", N);
    TC_REQUIRE(K % kt == 0, "tc_gemm: K=%d must be a multiple of %d", K, kt);
    TC_REQUIRE(lda >= K && ldo >= N, "tc_gemm: leading dimensions lda=%d ldo=%d too small", lda, ldo);
    TC_REQUIRE((lda * esz) % 16 == 0 && ldo % 8 == 0, "tc_gemm: leading dimensions must keep rows 16-byte aligned");
    TC_REQUIRE(tc_aligned16(A) && tc_aligned16(W) && tc_aligned16(out) && tc_aligned16(bias),
               "tc_gemm: operands must be 16-byte aligned");
    TC_REQUIRE(bias || epilogue == TC_EPI_PLAIN_F32 || epilogue == TC_EPI_DGELU || epilogue == TC_EPI_BIAS_RESIDUAL ||
                   epilogue == TC_EPI_BIAS,
               "tc_gemm: this epilogue needs a bias vector");
    TC_REQUIRE(splits >= 1 && splits <= 64, "tc_gemm: bad split count %d", splits);
    TC_REQUIRE(splits == 1 || epilogue == TC_EPI_PLAIN_F32, "tc_gemm: split-K needs the plain f32 epilogue");
    GemmParams p{};
    p.A = static_cast<const char *>(A);
    p.W = static_cast<const char *>(W);
    p.bias = bias;
    p.residual = residual;
    p.aux = static_cast<const char *>(aux);
    p.out = static_cast<char *>(out);
    p.out2 = static_cast<char *>(out2);
    p.rope_cos = rope_cos;
    p.rope_sin = rope_sin;
    p.pos = pos;
    p.lda_bytes = (long)lda * esz;
    p.ldw_bytes = (long)K * esz;
    p.ldo = ldo;
    p.M = M; p.N = N; p.K = K;
    p.nk = K / kt;
    p.T = T; p.head_dim = head_dim; p.E = N / 3;
    p.q_scale = q_scale;
    p.splits = splits;
    p.slab_elems = (long)M * ldo;
    if (epilogue == TC_EPI_BIAS_RESIDUAL) {
        TC_REQUIRE(residual && tc_aligned16(residual), "tc_gemm: residual epilogue needs an aligned residual");
    }
    if (epilogue == TC_EPI_BIAS_GELU_SAVE) TC_REQUIRE(out2 && tc_aligned16(out2), "tc_gemm: GELU_SAVE needs out2");
    if (epilogue == TC_EPI_DGELU) TC_REQUIRE(aux && tc_aligned16(aux), "tc_gemm: DGELU needs the pre-activation");
    if (epilogue == TC_EPI_QKV_ROPE) {
        TC_REQUIRE(N % 3 == 0 && (N / 3) % 128 == 0, "tc_gemm: qkv epilogue needs N = 3E with E %% 128 == 0");
        TC_REQUIRE(rope_cos && rope_sin && T > 0, "tc_gemm: qkv epilogue needs rope tables and T");
        TC_REQUIRE(head_dim > 0 && head_dim % 8 == 0 && (N / 3) % head_dim == 0,
                   "tc_gemm: head_dim=%d must be a multiple of 8 dividing E", head_dim);
        TC_REQUIRE(tc_aligned16(rope_cos) && tc_aligned16(rope_sin), "tc_gemm: rope tables must be 16-byte aligned");
    }
    if (g_num_cus == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) {
            tc_set_error("tc_gemm: cannot query the compute-unit count");
            return TC_ELAUNCH;
        }
        g_num_cus = cus;
    }
    // the large tile needs N % 256 == 0 (and, for the fused q/k/v epilogue, E % 256 == 0 so that
    // a tile never straddles two sections) and enough tiles to occupy every CU
    // (32-bit byte offsets: operands, and the result rows up to the end of the last tile)
    const int out_esz = (epilogue == TC_EPI_BIAS_RESIDUAL && !split_lo) || epilogue == TC_EPI_PLAIN_F32 ? 4 : esz;
    bool big = N % 256 == 0 && (long)(panels ? n_panels : (M + 255) / 256) * (N / 256) >= g_num_cus &&
               (long)M * lda * esz < (1L << 32) && (long)N * K * esz < (1L << 32) &&
               ((long)M + 256) * ldo * out_esz < (1L << 32);
    if (epilogue == TC_EPI_QKV_ROPE && (N / 3) % 256 != 0) big = false;
    if (dtype != TC_BF16) big = false;   // the f32 path is MFMA-rate bound (1/16 of bf16): small tile is enough
    if (splits > 1) big = false;
    {
        static int force = -2;
        if (force == -2) { const char *e = getenv("TC_GEMM_TILE"); force = e ? atoi(e) : -1; }
        if (force == 128) big = false;
    }
    if (split_lo) {
        p.res_lo = static_cast<const char *>(split_lo);
        p.out_lo = static_cast<char *>(split_lo);
    }
    if (ln_s || ln_rows || row_acc) {
        // the folded LayerNorm lives in the alternating large-tile kernel's epilogues only
        bool ok = big && alt_schedule_enabled();
        if (epilogue == TC_EPI_QKV_ROPE && ((T % 256 != 0 && !panels) || head_dim != 64 || pos)) ok = false;
        TC_REQUIRE(ok, "tc_gemm: the folded LayerNorm needs the large alternating kernel (M=%d N=%d K=%d T=%d)", M, N, K, T);
        p.ln_s = ln_s;
        p.ln_rows = ln_rows;
        p.row_acc = row_acc;
    }
    if (panels) {
        TC_REQUIRE(epilogue == TC_EPI_QKV_ROPE && ln_rows && big && n_panels > 0 && M >= 2 && tc_aligned16(panels),
                   "tc_gemm: a row panel list goes with the folded q/k/v product on the large alternating kernel");
        p.panels = panels;
        p.n_panels = n_panels;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    // DGELU: out2 (optional, f32 [N]) receives += column sums of the result.  The large tile adds
    // them in its epilogue; the other kernels are followed by a pass over the result.
    const bool want_colsum = epilogue == TC_EPI_DGELU && out2 != nullptr;
    if (epilogue == TC_EPI_DGELU) p.out2 = nullptr;
    p.colsum = want_colsum && big ? static_cast<float *>(out2) : nullptr;
    int rc;
    {
        TcProfScope prof(TC_PROF_GEMM, s, 2.0 * (double)M * (double)N * (double)K);
        rc = dtype == TC_BF16 ? dispatch<__bf16>(epilogue, p, big, s) : dispatch<float>(epilogue, p, big, s);
    }
    if (rc == TC_OK && want_colsum && !big)
        rc = tc_colsum(dtype, out, ldo, nullptr, 0, static_cast<float *>(out2), M, N, stream);
    return rc;
}

extern "C" int tc_gemm(int dtype, int epilogue, const void *A, int lda, const void *W, const float *bias,
                       const float *residual, void *out, int ldo, int M, int N, int K, float q_scale,
                       const float *rope_cos, const float *rope_sin, int T, int head_dim, void *stream) {
    TC_REQUIRE(epilogue >= TC_EPI_BIAS && epilogue <= TC_EPI_QKV_ROPE, "tc_gemm: unknown epilogue %d", epilogue);
    return gemm_impl(dtype, epilogue, A, lda, W, bias, residual, nullptr, out, nullptr, ldo, M, N, K, 1, q_scale,
                     rope_cos, rope_sin, T, head_dim, stream);
}

// TC_EPI_QKV_ROPE for packed batches: row m is rotated by row pos[m] of the tables (pos int32 [M] on
// the device, every entry below table_rows: the caller's contract, it is not checked on the device).
extern "C" int tc_gemm_qkv_packed(int dtype, const void *A, int lda, const void *W, const float *bias, void *out, int ldo,
                                  int M, int N, int K, float q_scale, const float *rope_cos, const float *rope_sin,
                                  const int32_t *pos, int table_rows, int head_dim, void *stream) {
    TC_REQUIRE(pos != nullptr && table_rows > 0, "tc_gemm_qkv_packed: needs the row positions and the table length");
    return gemm_impl(dtype, TC_EPI_QKV_ROPE, A, lda, W, bias, nullptr, nullptr, out, nullptr, ldo, M, N, K, 1, q_scale,
                     rope_cos, rope_sin, table_rows, head_dim, stream, pos);
}

// LayerNorm folded into the products on either side of it (see the header).
extern "C" int tc_gemm_ln(int dtype, int epilogue, const void *A, int lda, const void *Wf, const float *c,
                          const float *s, const float *rows, void *out, int ldo, int M, int N, int K, float q_scale,
                          const float *rope_cos, const float *rope_sin, int T, int head_dim, void *stream) {
    TC_REQUIRE(epilogue == TC_EPI_QKV_ROPE || epilogue == TC_EPI_BIAS_GELU, "tc_gemm_ln: epilogue %d has no folded form", epilogue);
    TC_REQUIRE(c && s && rows && tc_aligned16(s) && tc_aligned16(rows), "tc_gemm_ln: needs aligned c, s and row statistics");
    return gemm_impl(dtype, epilogue, A, lda, Wf, c, nullptr, nullptr, out, nullptr, ldo, M, N, K, 1, q_scale, rope_cos,
                     rope_sin, T, head_dim, stream, nullptr, s, rows, nullptr);
}

// The folded q/k/v product of a PACKED batch: row panel i of the launch is rows [panels[4i], panels[4i+1]) of A and
// out: at most 256 rows of one sequence whose rotary positions run from panels[4i+2] on (panels[4i+3] unused), so
// the table-slice rotary epilogue applies; a sequence of L rows contributes ceil(L / 256) panels, the last one
// short.  panels: int32 [n_panels, 4] on the device, rows and positions inside [0, M) and [0, table_rows): the
// caller's contract, not checked on the device.
extern "C" int tc_gemm_ln_qkv_panels(int dtype, const void *A, int lda, const void *Wf, const float *c, const float *s,
                                     const float *rows, void *out, int ldo, int M, int N, int K, float q_scale,
                                     const float *rope_cos, const float *rope_sin, int table_rows, int head_dim,
                                     const int32_t *panels, int n_panels, void *stream) {
    TC_REQUIRE(c && s && rows && tc_aligned16(s) && tc_aligned16(rows), "tc_gemm_ln_qkv_panels: needs aligned c, s and row statistics");
    TC_REQUIRE(panels && n_panels > 0 && table_rows >= 96, "tc_gemm_ln_qkv_panels: needs the panel list and rotary tables of 96 rows or more");
    return gemm_impl(dtype, TC_EPI_QKV_ROPE, A, lda, Wf, c, nullptr, nullptr, out, nullptr, ldo, M, N, K, 1, q_scale, rope_cos,
                     rope_sin, table_rows, head_dim, stream, nullptr, s, rows, nullptr, nullptr, panels, n_panels);
}

extern "C" int tc_gemm_residual_ln(int dtype, const void *A, int lda, const void *W, const float *bias,
                                   const float *residual, void *out, void *out_bf16, unsigned long long *row_acc,
                                   int ldo, int M, int N, int K, void *stream) {
    TC_REQUIRE(out_bf16 && row_acc && tc_aligned16(out_bf16) && tc_aligned16(row_acc),
               "tc_gemm_residual_ln: needs the bf16 copy and the row accumulators");
    return gemm_impl(dtype, TC_EPI_BIAS_RESIDUAL, A, lda, W, bias, residual, nullptr, out, out_bf16, ldo, M, N, K, 1, 0.f,
                     nullptr, nullptr, 0, 0, stream, nullptr, nullptr, nullptr, row_acc);
}

// The same with the residual stream held as two bf16 planes (x = hi + lo), updated in place: reads the
// planes as the residual, writes the new planes (the hi plane is what the next folded product multiplies)
// and the row sums of the new hi plane.  No f32 result exists.
extern "C" int tc_gemm_residual_split(int dtype, const void *A, int lda, const void *W, const float *bias, void *x_hi,
                                      void *x_lo, unsigned long long *row_acc, int ldo, int M, int N, int K,
                                      void *stream) {
    TC_REQUIRE(x_hi && x_lo && tc_aligned16(x_hi) && tc_aligned16(x_lo) && tc_aligned16(row_acc),
               "tc_gemm_residual_split: needs both planes (and row accumulators, when given), 16-byte aligned");
    return gemm_impl(dtype, TC_EPI_BIAS_RESIDUAL, A, lda, W, bias, static_cast<const float *>(x_hi), nullptr, x_hi, x_hi,
                     ldo, M, N, K, 1, 0.f, nullptr, nullptr, 0, 0, stream, nullptr, nullptr, nullptr, row_acc, x_lo);
}

// 1 when the products of a layer that carry no rotary epilogue (out-proj, fc1, fc2) take the alternating
// large-tile kernel for M rows, so that the second LayerNorm can be folded into fc1 and the residual stream
// kept as bf16 planes.
extern "C" int tc_ln_fold_rows_supported(int dtype, long M, int E, int F) {
    if (dtype != TC_BF16 || M <= 0 || !alt_schedule_enabled()) return 0;
    if (g_num_cus == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            return 0;
        g_num_cus = cus;
    }
    { const char *e = getenv("TC_GEMM_TILE"); if (e && atoi(e) == 128) return 0; }
    if (E % 256 != 0 || F % 256 != 0) return 0;
    if (((M + 255) / 256) * (E / 256) < g_num_cus) return 0;
    if (M * (F + 64) * 2 >= (1L << 32)) return 0;
    return 1;
}

// 1 when a forward of M = B*T rows can run with all its LayerNorms folded into the products: as above, and
// the q/k/v product's rotary epilogue takes its table-slice form (no sequence boundary inside a 256-row tile).
extern "C" int tc_ln_fold_supported(int dtype, int B, int T, int E, int F) {
    if (B <= 0 || T <= 0 || T % 256 != 0) return 0;
    return tc_ln_fold_rows_supported(dtype, (long)B * T, E, F);
}

extern "C" int tc_gemm_train(int dtype, int epilogue, const void *A, int lda, const void *W, const float *bias,
                             const void *aux, void *out, void *out2, int ldo, int M, int N, int K, int splits,
                             void *stream) {
    TC_REQUIRE(epilogue == TC_EPI_PLAIN_F32 || epilogue == TC_EPI_BIAS_GELU_SAVE || epilogue == TC_EPI_DGELU,
               "tc_gemm_train: unknown epilogue %d", epilogue);
    return gemm_impl(dtype, epilogue, A, lda, W, bias, nullptr, aux, out, out2, ldo, M, N, K, splits, 0.f, nullptr,
                     nullptr, 0, 0, stream);
}
