// Micro-benchmark: what one direct-to-LDS load costs the SIMD that issues it.
// A: one wave per SIMD runs 16 MFMAs per iteration with P load pieces (1 KiB each) spread between them.
// B: two waves per SIMD; waves 0-3 run the MFMAs only, waves 4-7 issue the P pieces only.
// Sources stay inside a 64 KiB window per workgroup (L2 hits), so this is issue cost, not bandwidth.
// Modes: 0 none, 1 global_load_lds_dwordx4, 2 raw_buffer_load_lds (dwordx4), 3 global_load_dwordx4 to VGPRs.
// Build: hipcc -O3 --offload-arch=gfx950 -o dmaissue dmaissue.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;

template <int MODE, int P, int SPLIT>
__global__ __launch_bounds__(SPLIT ? 512 : 256) void k(const char *src, int iters, unsigned long long *cyc, float *sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char *win = src + (size_t)blockIdx.x * 65536;
    const bool does_mfma = !SPLIT || wid < 4;
    const bool does_load = !SPLIT || wid >= 4;
    const int lw = wid & 3;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(lane + j); b[j] = (__bf16)(float)(lane - j); }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f32x4 vsum = {0, 0, 0, 0};
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)win, 0, 65536, 0x00020000);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const int slot = (it & 1) * 32768 + lw * 8192;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if (does_mfma) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
            if (does_load && P > 0 && (m % (16 / (P > 0 ? P : 1))) == 0) {
                const int piece = m / (16 / (P > 0 ? P : 1));
                const int goff = ((it * 8 + piece) & 7) * 1024 + lw * 8192 + lane * 16;
                char *l = smem + slot + (piece & 7) * 1024;
                if (MODE == 1)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(win + goff),
                                                     (__attribute__((address_space(3))) void *)l, 16, 0, 0);
                else if (MODE == 2)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)l, 16,
                                                         lane * 16, ((it * 8 + piece) & 7) * 1024 + lw * 8192, 0, 0);
                else if (MODE == 3)
                    vsum += *reinterpret_cast<const f32x4 *>(win + goff);
            }
        }
        if (does_load && MODE != 0 && MODE != 3) {
            if (P >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (P == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = vsum[0] + vsum[1] + vsum[2] + vsum[3];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 123.456f) sink[0] = s + smem[lane];
    if (lane == 0) cyc[blockIdx.x * 8 + wid] = t1 - t0;
}

template <int MODE, int P, int SPLIT>
void run(const char *name, const char *src, unsigned long long *cyc, float *sink) {
    const int iters = 2000, grid = 256, nw = SPLIT ? 8 : 4;
    CK(hipFuncSetAttribute((const void *)k<MODE, P, SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemset(cyc, 0, grid * 8 * sizeof(unsigned long long)));
        hipLaunchKernelGGL((k<MODE, P, SPLIT>), dim3(grid), dim3(64 * nw), 65536, 0, src, iters, cyc, sink);
        CK(hipDeviceSynchronize());
    }
    unsigned long long h[256 * 8];
    CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    double mf = 0, ld = 0;
    for (int b = 0; b < grid; ++b) for (int w = 0; w < nw; ++w) (w < 4 ? mf : ld) += (double)h[b * 8 + w];
    mf /= grid * 4.0 * iters;
    ld /= grid * 4.0 * iters;
    // s_memtime ticks at 100 MHz on this part; report ticks per iteration of the MFMA waves (and of the loading waves)
    printf("%-34s P=%d  mfma waves %8.3f ticks/iter", name, P, mf);
    if (SPLIT) printf("   load waves %8.3f", ld);
    printf("\n");
    fflush(stdout);
}

int main() {
    char *src;
    unsigned long long *cyc;
    float *sink;
    CK(hipMalloc(&src, (size_t)256 * 65536));
    CK(hipMemset(src, 1, (size_t)256 * 65536));
    CK(hipMalloc(&cyc, 256 * 8 * sizeof(unsigned long long)));
    CK(hipMalloc(&sink, 64));
    run<0, 0, 0>("A none", src, cyc, sink);
    run<1, 4, 0>("A global_load_lds", src, cyc, sink);
    run<1, 8, 0>("A global_load_lds", src, cyc, sink);
    run<2, 4, 0>("A buffer_load_lds", src, cyc, sink);
    run<2, 8, 0>("A buffer_load_lds", src, cyc, sink);
    run<3, 4, 0>("A global_load to VGPR", src, cyc, sink);
    run<3, 8, 0>("A global_load to VGPR", src, cyc, sink);
    run<0, 0, 1>("B none (partner idle loop)", src, cyc, sink);
    run<1, 8, 1>("B partner global_load_lds", src, cyc, sink);
    run<2, 8, 1>("B partner buffer_load_lds", src, cyc, sink);
    run<1, 16, 1>("B partner global_load_lds", src, cyc, sink);
    return 0;
}
