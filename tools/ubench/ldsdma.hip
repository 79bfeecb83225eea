// Micro-benchmark: per-CU global -> LDS (direct) and global -> VGPR load throughput.
// One 512-thread workgroup per CU; every wave issues PER_ITER 1-KiB loads per iteration.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int MODE, int DEPTH, int ROWB>   // MODE 0: glds, 1: vgpr loads; DEPTH batches in flight; ROWB bytes per row segment
__global__ __launch_bounds__(512) void k(const char* src, size_t span, int iters, unsigned long long* cyc, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // each wave's 8 loads of one iteration cover 8 KiB; the workgroup covers 64 KiB per iteration
    size_t base = ((size_t)blockIdx.x * 7919u * 65536u) % span;
    f32x4_dummy:;
    typedef __attribute__((ext_vector_type(4))) float f4;
    f4 acc = {0, 0, 0, 0};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        size_t off = (base + (size_t)it * 65536u) % span;
        const char* p = src + off + wid * 8192;
        char* l = smem + (it % DEPTH) * 65536 + wid * 8192;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // lane address inside a 1 KiB piece: rows of ROWB bytes taken from rows 4 KiB apart when ROWB < 1024
            const int per_row = ROWB / 16;
            const int r = lane / per_row, c = lane % per_row;
            const char* g = ROWB == 1024 ? p + i * 1024 + lane * 16 : p + (size_t)(i * (64 / per_row) + r) * ROWB + c * 16 + 0;
            if (MODE == 0) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(l + i * 1024), 16, 0, 0);
            } else {
                acc += *reinterpret_cast<const f4*>(g);
            }
        }
        if (MODE == 0) {
            if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (MODE == 1) sink[blockIdx.x * 512 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <int MODE, int DEPTH, int ROWB>
void run(const char* name, const char* src, size_t span, int grid, unsigned long long* cyc, float* sink) {
    const int iters = 2000;
    auto kern = k<MODE, DEPTH, ROWB>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 65536));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), DEPTH > 2 ? 2 * 65536 : DEPTH * 65536, 0, src, span, 50, cyc, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), DEPTH > 2 ? 2 * 65536 : DEPTH * 65536, 0, src, span, iters, cyc, sink);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[256]; CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    double avg = 0; for (int i = 0; i < grid; ++i) avg += h[i]; avg /= grid;
    double bytes = (double)iters * 65536;
    printf("%-34s grid %3d span %6zu MB: %7.1f GB/s per CU, %6.2f TB/s total, %5.2f B/clk/CU (clock %.2f GHz)\n", name, grid, span >> 20,
           bytes / (ms * 1e-3) / 1e9, bytes * grid / (ms * 1e-3) / 1e12, bytes / avg, avg / (ms * 1e-3) / 1e9);
}

int main() {
    size_t big = (size_t)2 << 30;
    char* src; CK(hipMalloc(&src, big)); CK(hipMemset(src, 1, big));
    unsigned long long* cyc; CK(hipMalloc(&cyc, 256 * 8));
    float* sink; CK(hipMalloc(&sink, 256 * 512 * 4));
    size_t l2 = (size_t)16 << 20, mall = (size_t)128 << 20;
    for (int grid : {256, 32}) {
        run<0, 1, 1024>("glds depth1 linear 1KB", src, l2, grid, cyc, sink);
        run<0, 2, 1024>("glds depth2 linear 1KB", src, l2, grid, cyc, sink);
        run<0, 1, 128>("glds depth1 rows128B", src, l2, grid, cyc, sink);
        run<0, 2, 128>("glds depth2 rows128B", src, l2, grid, cyc, sink);
        run<0, 2, 64>("glds depth2 rows64B", src, l2, grid, cyc, sink);
        run<1, 1, 1024>("vgpr loads linear", src, l2, grid, cyc, sink);
        run<0, 2, 128>("glds depth2 rows128B (MALL)", src, mall, grid, cyc, sink);
        run<0, 2, 128>("glds depth2 rows128B (HBM)", src, big, grid, cyc, sink);
        run<1, 1, 1024>("vgpr loads linear (HBM)", src, big, grid, cyc, sink);
    }
    return 0;
}
