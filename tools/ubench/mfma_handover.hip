// What does it cost to hand the matrix pipe of a SIMD from one wave to another, and to stop and restart it?
// One workgroup per CU.  Every variant issues the same number of mfma_f32_16x16x32_bf16 per SIMD (16 clocks each).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_handover mfma_handover.hip && ./mfma_handover
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int N>
__device__ __forceinline__ void burst(f32x4 (&acc)[16], const bf16x8 &a, const bf16x8 &b) {
#pragma unroll
    for (int m = 0; m < N; ++m) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m & 15]) : "v"(a), "v"(b));
}

// VARIANT
//  0: 4 waves (one per SIMD), 32 MFMAs per iteration, no barrier
//  1: 4 waves, 32 MFMAs, one barrier per iteration
//  2: 4 waves, 16 MFMAs + barrier, twice per iteration
//  3: 8 waves (two per SIMD), every wave 16 MFMAs per iteration, no barrier (free-running)
//  4: 8 waves, every wave 16 MFMAs, one barrier per iteration (both waves of a SIMD multiply at the same time)
//  5: 8 waves, the groups take turns: 16 MFMAs, barrier, the other group's 16 MFMAs, barrier
//  6: as 5 with 32-MFMA turns (counted as two iterations)
//  7: 8 waves, turns of 16 handed over through an LDS flag the waiting wave polls with s_sleep (no s_barrier)
//  8: 4 waves, 16 MFMAs then ~256 idle clocks (s_sleep), twice per iteration: does a pause cost more than its length?
//  10 + E: as 5, but a wave enters the hand-over barrier E MFMAs before the end of its turn and issues those after it,
//          beside the first MFMAs of the wave it released
template <int V>
__global__ __launch_bounds__(512) void k(float *out, int iters) {
    __shared__ volatile int flag[8];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(lane + j); b[j] = (__bf16)1.0f; }
    f32x4 acc[16] = {};
    if (threadIdx.x < 8) flag[threadIdx.x] = 0;
    __syncthreads();
    const int simd = wid & 3, grp = wid >> 2;
    if (V >= 10 && grp == 1) __builtin_amdgcn_s_barrier();
    for (int i = 0; i < iters; ++i) {
        if (V == 0) burst<32>(acc, a, b);
        if (V == 1) { burst<32>(acc, a, b); __builtin_amdgcn_s_barrier(); }
        if (V == 2) { burst<16>(acc, a, b); __builtin_amdgcn_s_barrier(); burst<16>(acc, a, b); __builtin_amdgcn_s_barrier(); }
        if (V == 3) burst<16>(acc, a, b);
        if (V == 4) { burst<16>(acc, a, b); __builtin_amdgcn_s_barrier(); }
        if (V == 5) {
            if (grp == 0) burst<16>(acc, a, b);
            __builtin_amdgcn_s_barrier();
            if (grp == 1) burst<16>(acc, a, b);
            __builtin_amdgcn_s_barrier();
        }
        if (V == 6) {
            if (grp == 0) burst<32>(acc, a, b);
            __builtin_amdgcn_s_barrier();
            if (grp == 1) burst<32>(acc, a, b);
            __builtin_amdgcn_s_barrier();
        }
        if (V == 7) {
            // flag[simd] counts finished turns; group g owns the turns with (count & 1) == g
            const int my_turn = 2 * i + grp;
            for (int spin = 0; flag[simd] != my_turn && spin < (1 << 20); ++spin) __builtin_amdgcn_s_sleep(1);   // bounded
            burst<16>(acc, a, b);
            if (lane == 0) flag[simd] = my_turn + 1;
        }
        if (V >= 10) {
            // one code path for both groups (group 1 entered one barrier before the loop): hipcc answers two
            // different instruction sequences around the same accumulators with copies of them
            constexpr int E = V - 10;
            burst<16 - E>(acc, a, b);
            __builtin_amdgcn_s_barrier();
            burst<E>(acc, a, b);
            __builtin_amdgcn_s_barrier();
        }
        if (V == 8) {
            burst<16>(acc, a, b);
            __builtin_amdgcn_s_sleep(4);
            burst<16>(acc, a, b);
            __builtin_amdgcn_s_sleep(4);
        }
    }
    if (V >= 10 && grp == 0) __builtin_amdgcn_s_barrier();
    float r = 0.f;
    for (int c = 0; c < 16; ++c) r += acc[c][c & 3];
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int V>
static void run(float *out, int threads, double mfma_per_simd_iter, const char *name) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 10000;
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<V>), dim3(256), dim3(threads), 0, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    printf("%-72s %7.3f ms  = %6.1f ns per 32 MFMAs of a SIMD\n", name, best, best * 1e6 / iters / (mfma_per_simd_iter / 32.0));
    hipEventDestroy(e0);
    hipEventDestroy(e1);
}

int main() {
    float *out;
    hipMalloc(&out, 256 * 512 * 4);
    run<0>(out, 256, 32, "one wave per SIMD, 32 MFMAs, no barrier");
    run<1>(out, 256, 32, "one wave per SIMD, 32 MFMAs + barrier");
    run<2>(out, 256, 32, "one wave per SIMD, (16 MFMAs + barrier) x 2");
    run<3>(out, 512, 32, "two waves per SIMD, 16 MFMAs each, free-running");
    run<4>(out, 512, 32, "two waves per SIMD, 16 MFMAs each + barrier (both multiply at once)");
    run<5>(out, 512, 32, "two waves per SIMD taking turns of 16 MFMAs, barrier after each turn");
    run<6>(out, 512, 64, "two waves per SIMD taking turns of 32 MFMAs, barrier after each turn");
    run<7>(out, 512, 32, "two waves per SIMD taking turns of 16 MFMAs through an LDS flag");
    run<8>(out, 256, 32, "one wave per SIMD, (16 MFMAs + s_sleep 4) x 2");
    run<10>(out, 512, 32, "turns of 16 on one code path, barrier at the end of the turn");
    run<11>(out, 512, 32, "turns of 16, barrier entered 1 MFMA early");
    run<12>(out, 512, 32, "turns of 16, barrier entered 2 MFMAs early");
    run<13>(out, 512, 32, "turns of 16, barrier entered 3 MFMAs early");
    run<14>(out, 512, 32, "turns of 16, barrier entered 4 MFMAs early");
    run<16>(out, 512, 32, "turns of 16, barrier entered 6 MFMAs early");
    return 0;
}
