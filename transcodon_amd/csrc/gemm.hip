// out[M,N] = epilogue(A[M,K] . W[N,K]^T + bias) on the gfx950 matrix cores.
//
// One 256-thread workgroup (4 waves, 2 x 2) owns a 128 x 128 output tile; every wave owns
// 64 x 64 of it as 4 x 4 MFMA 16x16 tiles.  Both operands are K-contiguous, so a k-tile is
// 128 rows x 128 BYTES for either element type (64 bf16 or 32 f32 per row), and the
// staging, swizzle and fragment addressing below are written in bytes and shared by both.
//
// Staging: global -> LDS by direct loads (global_load_lds_dwordx4): one wave instruction
// writes 1 KiB = 8 rows x 128 B linearly.  Bank conflicts of the later ds_read_b128
// fragment reads are avoided by an XOR swizzle of the 16-byte chunk index with (row & 7),
// applied on the per-lane SOURCE address and again on the read address.
// Two LDS stages: the loads of k-tile t+1 are in flight while tile t is multiplied.
//
// Epilogue: each wave parks its 64 x 64 f32 accumulators in its private LDS region and
// reads them back row-wise, 8 consecutive columns per lane, so that bias / GELU /
// residual / rotary arithmetic works on contiguous columns and global stores are 16 bytes.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128;
constexpr int TILE_BYTES = 128 * 128;          // one operand k-tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;    // A + B
constexpr int CST_LD = 68;                     // f32 per staged C row (64 + 4 pad)
constexpr int CST_WAVE_BYTES = 64 * CST_LD * 4;
constexpr int GEMM_LDS_BYTES = (4 * CST_WAVE_BYTES > 2 * STAGE_BYTES) ? 4 * CST_WAVE_BYTES
                                                                       : 2 * STAGE_BYTES;

struct GemmParams {
    const char *A;
    const char *W;
    const float *bias;
    const float *residual;
    char *out;
    const float *rope_cos;
    const float *rope_sin;
    long lda_bytes, ldw_bytes;
    int ldo;
    int M, N, K;
    int tiles_m, tiles_n;
    int nk;          // number of 128-byte k-tiles
    int T, head_dim, E;
    float q_scale;
};


__device__ __forceinline__ void glds16(const char *src, char *lds_wave_base) {
    __builtin_amdgcn_global_load_lds(TC_GLOBAL_PTR(src), TC_LDS_PTR(lds_wave_base), 16, 0, 0);
}

template <typename T>
__device__ __forceinline__ void mma_step(const f32x4 (&a)[4], const f32x4 (&b)[4],
                                         f32x4 (&acc)[4][4]);

// bf16: the 16 bytes of a fragment are 8 bf16 = one 16x16x32 operand.
template <>
__device__ __forceinline__ void mma_step<__bf16>(const f32x4 (&a)[4], const f32x4 (&b)[4],
                                                 f32x4 (&acc)[4][4]) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                __builtin_bit_cast(bf16x8, a[mi]), __builtin_bit_cast(bf16x8, b[ni]), acc[mi][ni],
                0, 0, 0);
}
// f32: the 16 bytes are 4 consecutive k of one row; MFMA j of the group of four takes
// element j from both operands, i.e. k-slot g of MFMA j stands for k = 4g + j in A and B alike.
template <>
__device__ __forceinline__ void mma_step<float>(const f32x4 (&a)[4], const f32x4 (&b)[4],
                                                f32x4 (&acc)[4][4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi][j], b[ni][j],
                                                                   acc[mi][ni], 0, 0, 0);
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

template <typename T, int EPI>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmParams p) {
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

    // ---- fragment read offsets (bytes inside an operand tile) ----
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

    stage(0, 0);
    for (int kt = 0; kt < p.nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                         // tile kt landed; buffer (kt+1)&1 is free
        if (kt + 1 < p.nk) stage((kt + 1) & 1, kt + 1);
        const char *sbase = smem + (kt & 1) * STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f32x4 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = *reinterpret_cast<const f32x4 *>(sbase + a_off[ks] + i * 16 * 128);
                b[i] = *reinterpret_cast<const f32x4 *>(sbase + b_off[ks] + i * 16 * 128);
            }
            mma_step<T>(a, b, acc);
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
    {
        const f32x4 b0 = *reinterpret_cast<const f32x4 *>(p.bias + n);
        const f32x4 b1 = *reinterpret_cast<const f32x4 *>(p.bias + n + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { bias[j] = b0[j]; bias[4 + j] = b1[j]; }
    }
    int section = 0, pair0 = 0;                  // QKV: 0 = q, 1 = k, 2 = v (uniform per block)
    if constexpr (EPI == TC_EPI_QKV_ROPE) {
        section = n0 / p.E;
        pair0 = (n % p.head_dim) >> 1;
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int row = it * 8 + rsub;
        const int m = m0 + wm * 64 + row;
        const f32x4 c0 = *reinterpret_cast<const f32x4 *>(cst + row * CST_LD + cg * 8);
        const f32x4 c1 = *reinterpret_cast<const f32x4 *>(cst + row * CST_LD + cg * 8 + 4);
        if (m >= p.M) continue;
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = c0[j] + bias[j]; v[4 + j] = c1[j] + bias[4 + j]; }
        const long o = (long)m * p.ldo + n;
        if constexpr (EPI == TC_EPI_BIAS) {
            store8<T>(p.out, o, v);
        } else if constexpr (EPI == TC_EPI_BIAS_GELU) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = tc_gelu(v[j]);
            store8<T>(p.out, o, v);
        } else if constexpr (EPI == TC_EPI_BIAS_RESIDUAL) {
            const f32x4 r0 = *reinterpret_cast<const f32x4 *>(p.residual + o);
            const f32x4 r1 = *reinterpret_cast<const f32x4 *>(p.residual + o + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = r0[j] + v[j]; v[4 + j] = r1[j] + v[4 + j]; }
            store8<float>(p.out, o, v);
        } else {  // TC_EPI_QKV_ROPE
            if (section == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] *= p.q_scale;
            }
            if (section < 2) {
                const int pos = m % p.T;
                const long t = (long)pos * (p.head_dim >> 1) + pair0;
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
}

template <typename T, int EPI>
int launch(const GemmParams &p, hipStream_t stream) {
    static bool attr_set = false;   // benign race: the attribute call is idempotent
    auto kern = gemm_kernel<T, EPI>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           GEMM_LDS_BYTES);
        if (e != hipSuccess) {
            tc_set_error("tc_gemm: hipFuncSetAttribute: %s", hipGetErrorString(e));
            return TC_ELAUNCH;
        }
        attr_set = true;
    }
    const int grid = p.tiles_m * p.tiles_n;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), GEMM_LDS_BYTES, stream, p);
    TC_CHECK_LAUNCH("tc_gemm");
    return TC_OK;
}

template <typename T>
int dispatch(int epilogue, const GemmParams &p, hipStream_t stream) {
    switch (epilogue) {
        case TC_EPI_BIAS: return launch<T, TC_EPI_BIAS>(p, stream);
        case TC_EPI_BIAS_GELU: return launch<T, TC_EPI_BIAS_GELU>(p, stream);
        case TC_EPI_BIAS_RESIDUAL: return launch<T, TC_EPI_BIAS_RESIDUAL>(p, stream);
        case TC_EPI_QKV_ROPE: return launch<T, TC_EPI_QKV_ROPE>(p, stream);
    }
    tc_set_error("tc_gemm: unknown epilogue %d", epilogue);
    return TC_EINVAL;
}

}  // namespace

extern "C" int tc_gemm(int dtype, int epilogue, const void *A, int lda, const void *W,
                       const float *bias, const float *residual, void *out, int ldo, int M, int N,
                       int K, float q_scale, const float *rope_cos, const float *rope_sin, int T,
                       int head_dim, void *stream) {
    TC_REQUIRE(dtype == TC_BF16 || dtype == TC_F32, "tc_gemm: bad dtype %d", dtype);
    const int esz = dtype == TC_BF16 ? 2 : 4;
    const int kt = 128 / esz;
    TC_REQUIRE(A && W && bias && out, "tc_gemm: null operand");
    TC_REQUIRE(M > 0 && N > 0 && K > 0, "tc_gemm: empty problem M=%d N=%d K=%d", M, N, K);
    TC_REQUIRE(N % BN == 0, "tc_gemm: N=%d must be a multiple of %d", N, BN);
    TC_REQUIRE(K % kt == 0, "tc_gemm: K=%d must be a multiple of %d", K, kt);
    TC_REQUIRE(lda >= K && ldo >= N, "tc_gemm: leading dimensions lda=%d ldo=%d too small", lda, ldo);
    TC_REQUIRE((lda * esz) % 16 == 0 && ldo % 8 == 0, "tc_gemm: leading dimensions must keep rows 16-byte aligned");
    TC_REQUIRE(tc_aligned16(A) && tc_aligned16(W) && tc_aligned16(out) && tc_aligned16(bias),
               "tc_gemm: operands must be 16-byte aligned");
    GemmParams p{};
    p.A = static_cast<const char *>(A);
    p.W = static_cast<const char *>(W);
    p.bias = bias;
    p.residual = residual;
    p.out = static_cast<char *>(out);
    p.rope_cos = rope_cos;
    p.rope_sin = rope_sin;
    p.lda_bytes = (long)lda * esz;
    p.ldw_bytes = (long)K * esz;
    p.ldo = ldo;
    p.M = M; p.N = N; p.K = K;
    p.tiles_m = (M + BM - 1) / BM;
    p.tiles_n = N / BN;
    p.nk = K / kt;
    p.T = T; p.head_dim = head_dim; p.E = N / 3;
    p.q_scale = q_scale;
    if (epilogue == TC_EPI_BIAS_RESIDUAL) {
        TC_REQUIRE(residual && tc_aligned16(residual), "tc_gemm: residual epilogue needs an aligned residual");
    }
    if (epilogue == TC_EPI_QKV_ROPE) {
        TC_REQUIRE(N % 3 == 0 && (N / 3) % BN == 0, "tc_gemm: qkv epilogue needs N = 3E with E %% %d == 0", BN);
        TC_REQUIRE(rope_cos && rope_sin && T > 0, "tc_gemm: qkv epilogue needs rope tables and T");
        TC_REQUIRE(head_dim > 0 && head_dim % 8 == 0 && (N / 3) % head_dim == 0,
                   "tc_gemm: head_dim=%d must be a multiple of 8 dividing E", head_dim);
        TC_REQUIRE(tc_aligned16(rope_cos) && tc_aligned16(rope_sin), "tc_gemm: rope tables must be 16-byte aligned");
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    TcProfScope prof(TC_PROF_GEMM, s, 2.0 * (double)M * (double)N * (double)K);
    return dtype == TC_BF16 ? dispatch<__bf16>(epilogue, p, s) : dispatch<float>(epilogue, p, s);
}
