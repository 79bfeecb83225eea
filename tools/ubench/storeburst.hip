// Micro-benchmark: the store pattern of the large-tile GEMM epilogue on its own.
// One 512-thread workgroup per CU walks TILES 256 x 256 output tiles; every wave writes its
// 128 x 64 (f32: 32 instructions of 16 B per lane) or packed bf16 part.  Variants:
//   pattern 0: four lanes write 64 contiguous bytes of a row, 16 rows per instruction (what the kernel does)
//   pattern 1: eight lanes write 128 contiguous bytes, 8 rows per instruction
//   pattern 2: 64 lanes write 1 KiB contiguous (one row segment): upper bound for the path
// gap: shader clocks of s_sleep-free spinning between tiles (emulates the k-loop), so that one
// can see whether a CU's burst or the chip's bandwidth sets the time.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef __attribute__((ext_vector_type(4))) float f4;

template <int PATTERN>
__global__ __launch_bounds__(512) void k(char* out, long ld_bytes, int tiles_n, int tiles, int gap, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    f4 v = {1.f * lane, 2.f, 3.f, 4.f};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < tiles; ++t) {
        const int tile = blockIdx.x + t * gridDim.x;
        const int tm = tile / tiles_n, tn = tile % tiles_n;
        char* base = out + (long)(tm * 256 + wr * 128) * ld_bytes + (long)(tn * 256 + wc * 64) * 4;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            char* p;
            if (PATTERN == 0) {          // instruction i: 16 rows x 64 B
                const int rb = i >> 2, cb = i & 3;
                p = base + (long)(rb * 16 + (lane >> 2)) * ld_bytes + cb * 64 + (lane & 3) * 16;
            } else if (PATTERN == 1) {   // 8 rows x 128 B
                const int rb = i >> 1, cb = i & 1;
                p = base + (long)(rb * 8 + (lane >> 3)) * ld_bytes + cb * 128 + (lane & 7) * 16;
            } else {                     // 4 rows x 256 B (the wave's whole 64 f32 columns)
                p = base + (long)(i * 4 + (lane >> 4)) * ld_bytes + (lane & 15) * 16;
            }
            *reinterpret_cast<f4*>(p) = v;
        }
        if (gap > 0) {
            const unsigned long long s = __builtin_amdgcn_s_memtime();
            while (__builtin_amdgcn_s_memtime() - s < (unsigned long long)gap) __builtin_amdgcn_s_sleep(8);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int PATTERN>
void run(const char* name, char* out, long ld_bytes, int M, int N, int gap, unsigned long long* cyc, int grid = 256) {
    const int tiles_n = N / 256, tiles = (M / 256) * tiles_n / 256, ntiles = tiles * grid;   // same tiles per workgroup
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<PATTERN>, dim3(grid), dim3(512), 0, 0, out, ld_bytes, tiles_n, tiles, gap, cyc);
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<PATTERN>, dim3(grid), dim3(512), 0, 0, out, ld_bytes, tiles_n, tiles, gap, cyc);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / 5, bytes = (double)ntiles * 256 * 256 * 4;
    printf("%-34s grid=%3d N=%d gap=%6d: %8.1f us  %6.2f TB/s (store-only time %8.1f us)\n", name, grid, N, gap, us, bytes / us / 1e6,
           us - (double)tiles * gap / 2300.0);
}

int main() {
    const int M = 131072;
    char* out; unsigned long long* cyc;
    CK(hipMalloc(&out, (size_t)M * 3200 * 4));
    CK(hipMalloc(&cyc, 256 * 8));
    for (int N : {768, 3072}) {
        for (int gap : {0, 40000}) {
            run<0>("4 lanes x 16 B per row, 16 rows", out, (long)N * 4, M, N, gap, cyc);
            run<1>("8 lanes x 16 B per row, 8 rows", out, (long)N * 4, M, N, gap, cyc);
            run<2>("16 lanes x 16 B per row, 4 rows", out, (long)N * 4, M, N, gap, cyc);
        }
    }
    // fewer workgroups, same work each: does one CU's burst drain faster when the others are idle?
    for (int grid : {256, 128, 64, 32, 8}) run<0>("4 lanes x 16 B per row, 16 rows", out, 3072L * 4, M, 3072, 0, cyc, grid);
    return 0;
}
