// Parameter block of the GEMM kernels of libtranscodon_hip.so (gemm.hip, gemm_w.hip): one struct, passed by
// value to every kernel; the host side (gemm.hip) fills it and picks the kernel.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

struct TcGemmParams {
    const char *A;
    const char *W;
    const float *bias;
    const float *residual;
    char *out;
    const float *rope_cos;
    const float *rope_sin;
    const int32_t *pos;    // TC_EPI_QKV_ROPE, optional: rotary position of every row (packed batches); else row % T
    const char *aux;       // TC_EPI_DGELU: pre-activation u, element type T, leading dimension ldo
    char *out2;            // TC_EPI_BIAS_GELU_SAVE: pre-activation output, element type T
    float *colsum;         // TC_EPI_DGELU, large tiles: column sums of the result are added here (f32 [N])
    // LayerNorm folded into the product (large alternating kernel only):
    //   consumer (TC_EPI_QKV_ROPE / TC_EPI_BIAS_GELU): W holds gamma o W, bias holds beta W^T + b, ln_s the
    //   row sums of the folded W, ln_rows per input row (rstd, -rstd * mean): result = rstd * acc +
    //   (-rstd * mean) * ln_s[n] + bias[n], then the epilogue proper;
    //   producer (TC_EPI_BIAS_RESIDUAL): out2 receives bf16(out) and row_acc += fixed-point (sum, sum of
    //   squares) of the bf16-rounded row segment of this tile (int64, 2^-24 units, [rows, 2]).
    const float *ln_s;
    const float *ln_rows;
    unsigned long long *row_acc;
    // producer in split form (gemm256x_kernel<bf16, TC_EPI_BIAS_RESIDUAL, FOLD>): the residual stream is a pair
    // of bf16 planes x = hi + lo (hi = bf16(x), lo = bf16(x - hi): 16 bits of mantissa); `residual` / `out2`
    // point to the hi plane, res_lo / out_lo to the lo plane; no f32 result is written
    const char *res_lo;
    char *out_lo;
    long lda_bytes, ldw_bytes;
    long slab_elems;       // split-K: f32 elements between the partial outputs of two splits
    int splits;            // split-K factor of the small-tile kernel (1 = off)
    int ldo;
    int M, N, K;
    int tiles_m, tiles_n;
    int nk;          // number of 128-byte k-tiles
    int T, head_dim, E;
    float q_scale;
    int gm;          // large-tile kernel: row panels per ordering group (1 = n-fastest)
    int gn;          // large-tile kernel: column panels per ordering group (tiles_n = no grouping)
    int cg_w0;       // alternating kernel: column panels of the first of two column groups (0 = one group)
    // Packed batches on the alternating q/k/v kernel: row panel i of the launch is rows [panels[4i], panels[4i+1])
    // (at most 256, all of one sequence, consecutive positions from panels[4i+2] on) instead of rows 256 i ..;
    // tiles_m = n_panels, T = rows of the rotary tables.
    const int32_t *panels;
    int n_panels;
    // measurement builds (-DTC_GEMM_STAMPS) only; always present so that every translation unit sees one layout
    unsigned long long *stamps;   // per workgroup: wait, issue, math, epilogue cycles (wave 0)
    int nostore;                  // experiment: the large-tile epilogue skips its final stores
};


// The one-wave-per-SIMD large-tile kernel (gemm_w.hip).  tc_gemm256w_supported: 1 when that kernel has the
// epilogue and takes the shape; tc_gemm256w_launch: launches it on `grid` persistent workgroups.
int tc_gemm256w_supported(int epilogue, const TcGemmParams &p);
int tc_gemm256w_launch(int epilogue, const TcGemmParams &p, int grid, hipStream_t stream);
