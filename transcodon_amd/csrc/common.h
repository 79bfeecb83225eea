// Internal helpers shared by the gfx950 kernels of libtranscodon_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/transcodon_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define TC_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void *)(p))
#define TC_LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))

#ifdef __HIPCC__
// A 1-KiB direct-to-LDS piece through a buffer descriptor (buffer_load_dwordx4 ... offen lds):
// the per-lane part of the address is a 32-bit VGPR that never changes, the moving part a scalar,
// so a piece costs no vector arithmetic at issue.  Raw buffer, stride 0, no bounds (callers clamp
// rows themselves); lane_off + uniform_off must stay below 4 GiB.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tc_buffer_rsrc(const void *base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, -1, 0x00020000);
}
__device__ __forceinline__ void tc_blds16(__amdgpu_buffer_rsrc_t rsrc, unsigned lane_off, unsigned uniform_off,
                                          char *lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, TC_LDS_PTR(lds_wave_base), 16, (int)lane_off, (int)uniform_off, 0, 0);
}
#endif

// ---- error reporting (host) ---------------------------------------------------------
void tc_set_error(const char *fmt, ...);

#define TC_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) {                        \
            tc_set_error(__VA_ARGS__);        \
            return TC_EINVAL;                 \
        }                                     \
    } while (0)

#define TC_CHECK_LAUNCH(what)                                                   \
    do {                                                                        \
        hipError_t e__ = hipGetLastError();                                     \
        if (e__ != hipSuccess) {                                                \
            tc_set_error("%s: %s", what, hipGetErrorString(e__));               \
            return TC_ELAUNCH;                                                  \
        }                                                                       \
    } while (0)

static inline bool tc_aligned16(const void *p) { return (((uintptr_t)p) & 15u) == 0; }

// ---- per-kernel-family event timing ---------------------------------------------------
// While profiling is enabled, a scope object names the kernel family and the work (flop or
// bytes) of the launches issued on this thread between its construction and destruction.  Each
// TC_LAUNCH inside it attaches a start and a stop event to the dispatch itself
// (hipExtLaunchKernelGGL): the timestamps come from the kernel's own packet, so timing adds no
// marker packets to the stream.  The scope's work and one launch count go to its first launch.
struct TcProfScope {
    int kind;
    double work;
    bool first, active;
    TcProfScope *outer;
    TcProfScope(int kind, hipStream_t stream, double work);
    ~TcProfScope();
};
// Events for the next launch of the innermost scope of this thread; false when there is none.
bool tc_prof_launch_events(hipEvent_t *start, hipEvent_t *stop);

#ifdef __HIPCC__
#include <hip/hip_ext.h>
#define TC_LAUNCH(kernel, grid, block, shmem, stream, ...)                                             \
    do {                                                                                               \
        hipEvent_t tc_e0__ = nullptr, tc_e1__ = nullptr;                                               \
        if (tc_prof_launch_events(&tc_e0__, &tc_e1__))                                                 \
            hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, tc_e0__, tc_e1__, 0, __VA_ARGS__); \
        else                                                                                           \
            hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                       \
    } while (0)
#endif

// ---- device helpers ---------------------------------------------------------------
// Sum over the 64 lanes on the VALU's lane crossbar (DPP) instead of six trips through the LDS
// permute network: quad, half-row and row mirrors leave every lane of a 16-lane row with the row
// sum, two row broadcasts fold the four rows into lane 63, and a lane read returns it to all.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float tc_dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float tc_wave_sum(float v) {
    v += tc_dpp_f32<0xB1, 0xf>(v);      // quad_perm [1,0,3,2]
    v += tc_dpp_f32<0x4E, 0xf>(v);      // quad_perm [2,3,0,1]
    v += tc_dpp_f32<0x141, 0xf>(v);     // row_half_mirror
    v += tc_dpp_f32<0x140, 0xf>(v);     // row_mirror
    v += tc_dpp_f32<0x142, 0xa>(v);     // row_bcast15 into rows 1 and 3
    v += tc_dpp_f32<0x143, 0xc>(v);     // row_bcast31 into rows 2 and 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float tc_wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
// exact erf GELU, x * 0.5 * (1 + erf(x / sqrt(2)))
__device__ __forceinline__ float tc_gelu(float x) {
    return x * 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ uint32_t tc_pack_bf16(float lo, float hi) {
    bf16x2 v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float tc_bf16_lo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float tc_bf16_hi(uint32_t w) {
    return __builtin_bit_cast(float, w & 0xffff0000u);
}

// XCD-aware remap of a linear workgroup id: workgroups that the dispatcher deals to the
// same XCD (id % 8 equal) get a contiguous range of logical ids, so neighbouring tiles
// share that XCD's L2.  Bijective for any grid size.
__device__ __forceinline__ int tc_xcd_remap(int id, int n) {
    const int q = n >> 3, r = n & 7;
    const int xcd = id & 7, k = id >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}
