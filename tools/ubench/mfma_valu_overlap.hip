// Do the matrix pipe and the vector ALU of one SIMD run side by side when the instructions come from
// DIFFERENT waves?  One workgroup of 8 waves per CU (two waves per SIMD: waves w and w + 4 share one).
//   waves 0-3 issue MFMAs (or idle), waves 4-7 issue VALU work (or idle); timed alone and together.
// If the pipes overlap, time(both) = max(time(mfma), time(valu)); if they share the issue slot, the sum.
// Variants: MFMA shape (32x32x16: 32 clocks each, 16x16x32: 16 clocks each), an `s_nop` pad after every
// MFMA (so that the multiplying wave does not sit in the issue stage with its next MFMA while the pipe is
// busy), and a raised priority for the VALU waves.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int PAD>
__device__ __forceinline__ void pad() {
    if constexpr (PAD == 1) asm volatile("s_nop 3");
    if constexpr (PAD == 2) asm volatile("s_nop 5");
    if constexpr (PAD == 3) asm volatile("s_nop 1");
}

// SHAPE 0: 4 x 32x32x16 per iteration (128 clocks); SHAPE 1: 8 x 16x16x32 per iteration (128 clocks)
template <int SHAPE, int PAD, int VALU_KIND, int PRIO>
__global__ __launch_bounds__(512) void k(float *out, int iters, int do_mfma, int do_valu) {
    const int wid = threadIdx.x >> 6;
    const bool mfma_wave = wid < 4;
    float r = 0.f;
    if (mfma_wave && do_mfma) {
        bf16x8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x + j); b[j] = (__bf16)1.0f; }
        if constexpr (SHAPE == 0) {
            f32x16 acc[4] = {};
            for (int i = 0; i < iters; ++i)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
                    pad<PAD>();
                }
            r = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
        } else {
            f32x4 acc[8] = {};
            for (int i = 0; i < iters; ++i)
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[c], 0, 0, 0);
                    pad<PAD>();
                }
            for (int c = 0; c < 8; ++c) r += acc[c][c & 3];
        }
    } else if (!mfma_wave && do_valu) {
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(3);
        float v[8];
        if constexpr (VALU_KIND == 0) {
            for (int j = 0; j < 8; ++j) v[j] = (float)(threadIdx.x + j);
            for (int i = 0; i < iters; ++i)
#pragma unroll
                for (int rep = 0; rep < 4; ++rep)
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = __builtin_fmaf(v[j], 1.0001f, 0.5f);   // 32 VALU per iteration: 128 clocks
        } else {
            for (int j = 0; j < 8; ++j) v[j] = (float)(threadIdx.x + j) * 1e-3f;
            for (int i = 0; i < iters; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = __builtin_amdgcn_exp2f(v[j]) * 0.25f;        // 8 exp + 8 mul per iteration
        }
        for (int j = 0; j < 8; ++j) r += v[j];
    }
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

// Same-wave variant: every active wave issues 4 x 32x32x16 per iteration with FILL independent v_fma_f32 after
// each MFMA (order pinned by scheduling barriers).  WAVES = 4: one wave per SIMD, 8: two.
template <int FILL, int WAVES, bool MFMA, int KIND>
__global__ __launch_bounds__(512) void k_same(float *out, int iters) {
    const int wid = threadIdx.x >> 6;
    float r = 0.f;
    if (wid < WAVES) {
        bf16x8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x + j); b[j] = (__bf16)1.0f; }
        f32x16 acc[4] = {};
        float v[8];
        for (int j = 0; j < 8; ++j) v[j] = (float)(threadIdx.x + j);
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                // volatile assembly keeps the written order (the compiler would sink and pack the fillers)
                if constexpr (MFMA)
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b));
#pragma unroll
                for (int f = 0; f < FILL; ++f) {
                    if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, 0.5" : "+v"(v[f & 7]) : "s"(1.0001f));
                    else asm volatile("v_exp_f32 %0, %0" : "+v"(v[f & 7]));
                }
            }
        r = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
        for (int j = 0; j < 8; ++j) r += v[j];
    }
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int FILL, int WAVES, int KIND = 0>
static void run_same(float *out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float t[2];
    for (int m = 0; m < 2; ++m) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (m == 0) hipLaunchKernelGGL((k_same<FILL, WAVES, true, KIND>), dim3(256), dim3(512), 0, 0, out, 20000);
            else hipLaunchKernelGGL((k_same<FILL, WAVES, false, KIND>), dim3(256), dim3(512), 0, 0, out, 20000);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        t[m] = best;
    }
    printf("same wave, %d waves/SIMD, %2d %s per 32x32x16:  with MFMA %7.3f ms, fillers only %7.3f ms\n", WAVES / 4, FILL, KIND ? "exp" : "fma", t[0], t[1]);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
}

template <int SHAPE, int PAD, int VALU_KIND, int PRIO>
static void run(float *out, const char *name) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    float t[3];
    const int cfg[3][2] = {{1, 0}, {0, 1}, {1, 1}};
    for (int m = 0; m < 3; ++m) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((k<SHAPE, PAD, VALU_KIND, PRIO>), dim3(256), dim3(512), 0, 0, out, iters, cfg[m][0], cfg[m][1]);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        t[m] = best;
    }
    printf("%-44s mfma %7.3f  valu %7.3f  both %7.3f ms  (sum %.3f, max %.3f)\n", name, t[0], t[1], t[2], t[0] + t[1],
           t[0] > t[1] ? t[0] : t[1]);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
}

int main() {
    float *out;
    hipMalloc(&out, 256 * 512 * 4);
    run<0, 0, 0, 0>(out, "32x32x16, fma");
    run<0, 0, 1, 0>(out, "32x32x16, exp");
    run<0, 3, 0, 0>(out, "32x32x16 + s_nop 1, fma");
    run<0, 1, 0, 0>(out, "32x32x16 + s_nop 3, fma");
    run<0, 2, 0, 0>(out, "32x32x16 + s_nop 5, fma");
    run<0, 0, 0, 1>(out, "32x32x16, fma at priority 3");
    run<0, 1, 0, 1>(out, "32x32x16 + s_nop 3, fma at priority 3");
    run<1, 0, 0, 0>(out, "16x16x32, fma");
    run<1, 3, 0, 0>(out, "16x16x32 + s_nop 1, fma");
    run<1, 0, 0, 1>(out, "16x16x32, fma at priority 3");
    run<1, 3, 0, 1>(out, "16x16x32 + s_nop 1, fma at priority 3");
    run_same<0, 4>(out);
    run_same<4, 4>(out);
    run_same<6, 4>(out);
    run_same<8, 4>(out);
    run_same<12, 4>(out);
    run_same<0, 8>(out);
    run_same<2, 8>(out);
    run_same<3, 8>(out);
    run_same<4, 8>(out);
    run_same<6, 8>(out);
    run_same<8, 8>(out);
    run_same<2, 4, 1>(out);
    run_same<3, 4, 1>(out);
    run_same<4, 4, 1>(out);
    run_same<2, 8, 1>(out);
    return 0;
}
