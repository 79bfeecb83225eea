// Do LDS fragment reads and direct-to-LDS loads of one SIMD hide under its MFMAs, and does it matter
// whether they are issued by the multiplying wave itself or by the other wave of the SIMD?
// One workgroup of 8 waves per CU (waves w and w + 4 share a SIMD).  Per iteration and SIMD the work is always
// 32 x mfma_f32_16x16x32_bf16 (512 clocks of matrix pipe), R ds_read_b128 and D direct-to-LDS pieces (1 KiB each):
//   CROSS: waves 0-3 issue the 32 MFMAs, waves 4-7 the R reads and D pieces (the alternating-group GEMM's split)
//   SAME:  every wave issues 16 MFMAs with R/2 reads and D/2 pieces spread between them
//   hipcc --offload-arch=gfx950 -O3 -o mfma_load_overlap mfma_load_overlap.hip && ./mfma_load_overlap
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, 0xffffffff, 0x00020000);
}

// MODE 0: CROSS, 1: SAME.  DO_MFMA / R / D select the work.  (Turn-taking schedules are in mfma_handover.hip: they
// need one code path for both wave groups, or hipcc copies the accumulators at every branch.)
template <int MODE, bool DO_MFMA, int R, int D>
__global__ __launch_bounds__(512) void k(float *out, const char *src, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(src + (size_t)blockIdx.x * 65536);
    char *my_lds = smem + wid * 16384;                  // 8 x 16 KiB = 128 KiB
    const unsigned rd = (unsigned)(unsigned long)LDS_PTR(my_lds) + lane * 16;   // conflict-free b128 reads
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(lane + j); b[j] = (__bf16)1.0f; }
    f32x4 acc[16] = {};
    u32x4 frag[4] = {};
    float r = 0.f;
    constexpr int kMfma = MODE == 0 ? 32 : 16;          // per wave and iteration
    constexpr int kR = MODE == 0 ? R : R / 2;
    constexpr int kD = MODE == 0 ? D : D / 2;
    const bool mfma_wave = MODE == 1 || wid < 4;
    const bool load_wave = MODE == 1 || wid >= 4;
    for (int i = 0; i < iters; ++i) {
        const unsigned goff = (unsigned)((i & 7) * 8192);
        if (MODE == 1) {
            int ri = 0, di = 0;
#pragma unroll
            for (int m = 0; m < kMfma; ++m) {
                if (DO_MFMA) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m & 15]) : "v"(a), "v"(b));
                // spread: reads after MFMA m when m * kR / kMfma advances, pieces likewise
                if (kR > 0 && (m * kR) / kMfma != ((m + 1) * kR) / kMfma) {
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(frag[ri & 3]) : "v"(rd), "n"(0));
                    ++ri;
                }
                if (kD > 0 && (m * kD) / kMfma != ((m + 1) * kD) / kMfma) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(my_lds + 1024 * (di & 7) + 4096), 16, lane * 16, goff + di * 1024, 0, 0);
                    ++di;
                }
            }
        } else {
            if (mfma_wave) {
#pragma unroll
                for (int m = 0; m < kMfma; ++m)
                    if (DO_MFMA) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m & 15]) : "v"(a), "v"(b));
            }
            if (load_wave) {
#pragma unroll
                for (int di = 0; di < kD; ++di)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(my_lds + 1024 * (di & 7) + 4096), 16, lane * 16, goff + di * 1024, 0, 0);
#pragma unroll
                for (int ri = 0; ri < kR; ++ri)
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(frag[ri & 3]) : "v"(rd), "n"(0));
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(frag[0]), "+v"(frag[1]), "+v"(frag[2]), "+v"(frag[3])::"memory");
        __builtin_amdgcn_s_barrier();                   // the GEMM's phases end in a barrier too
    }
    for (int c = 0; c < 16; ++c) r += acc[c][c & 3];
    for (int c = 0; c < 4; ++c) r += (float)frag[c][0];
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int MODE, bool DO_MFMA, int R, int D>
static float run(float *out, const char *src) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE, DO_MFMA, R, D>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, DO_MFMA, R, D>), dim3(256), dim3(512), 131072, 0, out, src, 10000);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return best;
}

template <int R, int D>
static void row(float *out, const char *src) {
    const float m0 = run<0, true, 0, 0>(out, src);
    const float lc = run<0, false, R, D>(out, src), ls = run<1, false, R, D>(out, src);
    const float bc = run<0, true, R, D>(out, src), bs = run<1, true, R, D>(out, src);
    printf("R=%2d reads D=%2d pieces per 32 MFMAs:  MFMA only %6.3f | loads only: cross %6.3f same %6.3f | both: cross %6.3f same %6.3f ms\n",
           R, D, m0, lc, ls, bc, bs);
}

int main() {
    float *out;
    char *src;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&src, (size_t)256 * 65536 + 65536);
    hipMemset(src, 1, (size_t)256 * 65536 + 65536);
    row<16, 0>(out, src);
    row<32, 0>(out, src);
    row<0, 4>(out, src);
    row<0, 8>(out, src);
    row<12, 4>(out, src);
    row<16, 4>(out, src);
    row<24, 4>(out, src);
    return 0;
}
