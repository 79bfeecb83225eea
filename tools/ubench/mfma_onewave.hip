// One wave per SIMD (256-thread workgroup, one per CU): does a wave's own stream of LDS fragment reads and
// direct-to-LDS pieces hide under its MFMAs?  Per iteration and wave: 128 x mfma_f32_16x16x32_bf16 (2048 clocks of
// matrix pipe: the k-tile of a 128 x 128 per-wave block), R ds_read_b128 and D direct-to-LDS pieces (1 KiB each)
// spread evenly between the MFMAs, one barrier.  For comparison the same per-CU work on 8 waves (two per SIMD, each
// 64 MFMAs, R/2 reads... per wave would be the alternating kernel; see mfma_load_overlap.hip).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_onewave mfma_onewave.hip && ./mfma_onewave
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, 0xffffffff, 0x00020000);
}

// WAIT: 0 = one vmcnt(0) + lgkmcnt(0) + barrier per iteration; 1 = counted vmcnt(D) (the pieces of this iteration stay
// in flight over the barrier, those of the one before are waited for)
template <bool DO_MFMA, int R, int D, int WAIT>
__global__ __launch_bounds__(256) void k(float *out, const char *src, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(src + (size_t)blockIdx.x * 131072);
    char *my_lds = smem + wid * 32768;                  // 4 x 32 KiB = 128 KiB
    const unsigned rd = (unsigned)(unsigned long)LDS_PTR(my_lds) + lane * 16;   // conflict-free b128 reads
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(lane + j); b[j] = (__bf16)1.0f; }
    f32x4 acc[64] = {};
    u32x4 frag[8] = {};
    float r = 0.f;
    for (int i = 0; i < iters; ++i) {
        const unsigned goff = (unsigned)((i & 3) * 32768);
        int ri = 0, di = 0;
#pragma unroll
        for (int m = 0; m < 128; ++m) {
            if (DO_MFMA) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[m & 63]) : "v"(a), "v"(b));
            if (R > 0 && (m * R) / 128 != ((m + 1) * R) / 128) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(frag[ri & 7]) : "v"(rd), "n"(0));
                ++ri;
            }
            if (D > 0 && (m * D) / 128 != ((m + 1) * D) / 128) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(my_lds + 1024 * (di & 15) + 8192), 16, lane * 16, goff + di * 1024, 0, 0);
                ++di;
            }
        }
        if (WAIT == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(D) : "memory");
        asm volatile("" : "+v"(frag[0]), "+v"(frag[1]), "+v"(frag[2]), "+v"(frag[3]), "+v"(frag[4]), "+v"(frag[5]), "+v"(frag[6]), "+v"(frag[7]));
        __builtin_amdgcn_s_barrier();
    }
#pragma unroll
    for (int c = 0; c < 64; ++c) r += acc[c][c & 3];
#pragma unroll
    for (int c = 0; c < 8; ++c) r += (float)frag[c][0];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <bool DO_MFMA, int R, int D, int WAIT>
static float run(float *out, const char *src) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<DO_MFMA, R, D, WAIT>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<DO_MFMA, R, D, WAIT>), dim3(256), dim3(256), 131072, 0, out, src, 4000);
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
static void row(float *out, const char *src, float m0) {
    const float l0 = run<false, R, D, 0>(out, src), b0 = run<true, R, D, 0>(out, src), b1 = run<true, R, D, 1>(out, src);
    printf("R=%2d reads D=%2d pieces per 128 MFMAs: loads only %6.3f | both, drained each iteration %6.3f (%.3f of MFMA only) | "
           "both, pieces one iteration in flight %6.3f (%.3f) ms\n", R, D, l0, b0, b0 / m0, b1, b1 / m0);
}

int main() {
    float *out;
    char *src;
    hipMalloc(&out, 256 * 256 * 4);
    hipMalloc(&src, (size_t)256 * 131072 + 131072);
    hipMemset(src, 1, (size_t)256 * 131072 + 131072);
    const float m0 = run<true, 0, 0, 0>(out, src);
    printf("MFMA only: %6.3f ms for 4000 iterations of 128 MFMAs per wave (%.0f ns per iteration)\n", m0, m0 * 1e6 / 4000);
    row<32, 0>(out, src, m0);
    row<0, 8>(out, src, m0);
    row<0, 16>(out, src, m0);
    row<32, 16>(out, src, m0);
    row<48, 16>(out, src, m0);
    row<32, 24>(out, src, m0);
    return 0;
}
