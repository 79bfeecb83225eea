// 256 x 256 tile with ONE wave per SIMD: four waves of 128 x 128 each.
//
// The alternating-group kernel of gemm.hip (eight waves of 64 x 32 in each quarter) reads 192 KiB of fragments
// out of LDS and issues 64 direct-to-LDS pieces per k-tile from eight instruction streams, and its k-tile lasts as
// long as the load slots of the wave that is not multiplying (3.0 k clocks for 2.05 k clocks of matrix work).  Here a
// wave owns a quarter of the tile: 128 KiB of fragment reads per k-tile, 256 accumulators per lane in the
// accumulation registers, and its own stream carries everything: per k-tile and wave 128 MFMAs
// (mfma_f32_16x16x32_bf16), 32 ds_read_b128 and 16 direct-to-LDS pieces, one LDS instruction and (in the second
// half) one piece after every fourth MFMA.  A wave that issues a read or a piece right behind an MFMA loses no
// matrix time while the matrix pipe works off that MFMA; measured on this mix: 1.07 x the bare MFMA time.
//
// k-tile g lives in LDS buffer g & 1 (64 KiB: half-tiles A0 A1 B0 B1 in the layout and swizzles of the alternating
// kernel).  First half of a k-tile: the MFMAs of k-step 0, between them the fragment reads of k-step 1.  Then the one
// wait + barrier of the k-tile: k-tile g + 1 has landed (its pieces were issued a k-tile ago) and every wave has
// read the last of buffer g & 1.  Second half: the MFMAs of k-step 1, between them the fragment reads of k-step 0 of
// k-tile g + 1 and the pieces of k-tile g + 2 into the buffer just freed.
#include "common.h"
#include "gemm_params.h"
#include <type_traits>
#include <stdlib.h>

namespace {
using GemmParams = TcGemmParams;

// measurement builds (-DTC_GEMM_STAMPS): wave 0 of every workgroup adds up the clocks of the k-tile's parts
#ifdef TC_GEMM_STAMPS
#define TC_WSTAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define TC_WADD(slot, a, b) c_st[slot] += (b) - (a)
#else
#define TC_WSTAMP(var)
#define TC_WADD(slot, a, b)
#endif

namespace kw {
constexpr int BM = 256, BN = 256;
constexpr int HALF_BYTES = 128 * 128;           // one half-tile (16 KiB): 128 rows x 128 bytes
constexpr int DBUF_BYTES = 4 * HALF_BYTES;      // A0 A1 B0 B1
constexpr int SLOT_A0 = 0, SLOT_B0 = 2 * HALF_BYTES;
constexpr int BIAS_OFF = 2 * DBUF_BYTES;        // per wave 1 KiB (one piece): the bias of its 128 columns in the first half
constexpr int LDS_BYTES = BIAS_OFF + 4 * 1024;
}  // namespace kw

__device__ __forceinline__ __amdgpu_buffer_rsrc_t bounded_rsrc(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}

template <int EPI>
__global__ __launch_bounds__(256) void gemm256w_kernel(const GemmParams p) {
    using namespace kw;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wid >> 1, wc = wid & 1;                  // the wave's quarter: rows wr*128.., columns wc*128..
    const int frow = lane & 15, fgrp = lane >> 4;

    // ---- tile ownership and order: as in the alternating kernel (gemm.hip) ----
    const int ntiles = p.tiles_m * p.tiles_n;
    const int cls = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int per_class = (gridDim.x + 7 - cls) >> 3;
    const int q = ntiles >> 3, r = ntiles & 7;
    const int first = cls < r ? cls * (q + 1) : r * (q + 1) + (cls - r) * q;
    const int count = q + (cls < r ? 1 : 0);
    const int my_count = count > j ? (count - j + per_class - 1) / per_class : 0;
    if (my_count == 0) return;
    const int w0 = p.cg_w0 > 0 && p.cg_w0 < p.tiles_n ? p.cg_w0 : p.tiles_n;
    const int w1 = p.tiles_n - w0, w1s = w1 > 0 ? w1 : 1;
    const int base1 = p.tiles_m * w0;
    const int sm0 = per_class / w0, sn0 = per_class - sm0 * w0;
    const int sm1 = per_class / w1s, sn1 = per_class - sm1 * w1s;
    auto locate = [&](int idx, int &tm, int &tn) {
        if (idx < base1) { tm = idx / w0; tn = idx - tm * w0; }
        else { const int l = idx - base1; tm = l / w1s; tn = w0 + l - tm * w1s; }
    };
    const int tile0 = first + j;
    int tm0, tn0;
    locate(tile0, tm0, tn0);
    auto advance = [&](int &tm, int &tn, int &idx) {
        const int nidx = idx + per_class;
        if (idx < base1) {
            if (nidx >= base1) locate(nidx, tm, tn);
            else { tn += sn0; tm += sm0; if (tn >= w0) { tn -= w0; ++tm; } }
        } else {
            tn += sn1; tm += sm1;
            if (tn >= p.tiles_n) { tn -= w1s; ++tm; }
        }
        idx = nidx;
    };

    // ---- staging: a half-tile is 16 pieces of 8 rows x 128 B; this wave moves pieces 4 wid .. 4 wid + 3 of each
    // of the four half-tiles: 16 pieces per k-tile.  Bounded descriptors: rows past the end of A read as zero.
    const __amdgpu_buffer_rsrc_t rsrc_a = bounded_rsrc(p.A, (unsigned)((long)p.M * p.lda_bytes));
    const __amdgpu_buffer_rsrc_t rsrc_w = bounded_rsrc(p.W, (unsigned)((long)p.N * p.ldw_bytes));
    const int srow = lane >> 3;
    const unsigned a_voff = (unsigned)srow * (unsigned)p.lda_bytes + (unsigned)(((lane & 7) ^ srow) * 16);
    // W rows, interleaved column order: key = (r & 3) | ((r >> 1) & 4); bit 3 of the row is the piece's parity
    const unsigned b_voff0 = (unsigned)srow * (unsigned)p.ldw_bytes + (unsigned)(((lane & 7) ^ (srow & 3)) * 16);
    const unsigned b_voff1 = (unsigned)srow * (unsigned)p.ldw_bytes + (unsigned)(((lane & 7) ^ ((srow & 3) | 4)) * 16);
    const unsigned lda8 = (unsigned)p.lda_bytes * 8u, ldw8 = (unsigned)p.ldw_bytes * 8u;
    int cm = tm0, cn = tn0, ci = tile0;          // staging cursor: tile, its index, k-tile, tiles done
    int c_tile = 0, c_k = 0;
    unsigned a_soff = 0, b_soff = 0;             // scalar byte offsets of the cursor's tile: this wave's first piece
    auto set_cursor = [&]() {
        a_soff = (unsigned)(cm * BM + 32 * wid) * (unsigned)p.lda_bytes;
        b_soff = (unsigned)(cn * BN + 32 * wid) * (unsigned)p.ldw_bytes;
    };
    set_cursor();
    // piece i (0..15) of the cursor's k-tile into buffer `buf`: i >> 3 = operand, (i >> 2) & 1 = half, i & 3 = piece
    auto stage_piece = [&](int buf_off, auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value, h = (i >> 2) & 1, jj = i & 3;
        char *dst = smem + buf_off + (i >> 3) * 2 * HALF_BYTES + h * HALF_BYTES + (4 * wid + jj) * 1024;
        const unsigned koff = (unsigned)c_k * 128u;
        if constexpr ((i >> 3) == 0)
            tc_blds16(rsrc_a, a_voff, a_soff + (unsigned)(h * 16 + jj) * lda8 + koff, dst);
        else
            tc_blds16(rsrc_w, (jj & 1) ? b_voff1 : b_voff0, b_soff + (unsigned)(h * 16 + jj) * ldw8 + koff, dst);
    };
    auto cursor_next = [&]() {
        if (c_tile >= my_count) return;
        if (c_k + 1 < p.nk) { ++c_k; return; }
        if (++c_tile < my_count) { c_k = 0; advance(cm, cn, ci); set_cursor(); }
    };
    auto stage_all = [&](int buf_off) __attribute__((always_inline)) {
        stage_piece(buf_off, std::integral_constant<int, 0>{});  stage_piece(buf_off, std::integral_constant<int, 1>{});
        stage_piece(buf_off, std::integral_constant<int, 2>{});  stage_piece(buf_off, std::integral_constant<int, 3>{});
        stage_piece(buf_off, std::integral_constant<int, 4>{});  stage_piece(buf_off, std::integral_constant<int, 5>{});
        stage_piece(buf_off, std::integral_constant<int, 6>{});  stage_piece(buf_off, std::integral_constant<int, 7>{});
        stage_piece(buf_off, std::integral_constant<int, 8>{});  stage_piece(buf_off, std::integral_constant<int, 9>{});
        stage_piece(buf_off, std::integral_constant<int, 10>{}); stage_piece(buf_off, std::integral_constant<int, 11>{});
        stage_piece(buf_off, std::integral_constant<int, 12>{}); stage_piece(buf_off, std::integral_constant<int, 13>{});
        stage_piece(buf_off, std::integral_constant<int, 14>{}); stage_piece(buf_off, std::integral_constant<int, 15>{});
    };

    // ---- fragment addresses (byte offsets inside a buffer) ----
    // A{wr}: row mi*16 + frow, chunk (ks*4 + fgrp) ^ (row & 7).  B{wc}: 32-column block cb, accumulator tile nb of
    // the pair, operand row frow -> half-tile row cb*32 + (frow >> 2)*8 + nb*4 + (frow & 3), key as staged.
    int a_off[2], b_off[2];
    {
        const int brow = (frow >> 2) * 8 + (frow & 3);
        const int key = (frow & 3) | (((frow >> 2) & 1) << 2);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            a_off[ks] = SLOT_A0 + wr * HALF_BYTES + frow * 128 + (((ks * 4 + fgrp) ^ (frow & 7)) << 4);
            b_off[ks] = SLOT_B0 + wc * HALF_BYTES + brow * 128 + (((ks * 4 + fgrp) ^ key) << 4);
        }
    }
    f32x4 fa[2][8], fb[2][8];                    // [k-step][row block] and [k-step][accumulator column tile]
    // fragment i (0..15) of k-step ks from the buffer at `buf`: i < 8: fb[ks][i] (block i >> 1, tile i & 1), else fa
    auto read_frag = [&](const char *buf, int ks, auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        if constexpr (i < 8)
            fb[ks][i] = *reinterpret_cast<const f32x4 *>(buf + b_off[ks] + (i >> 1) * 4096 + (i & 1) * 512);
        else
            fa[ks][i - 8] = *reinterpret_cast<const f32x4 *>(buf + a_off[ks] + (i - 8) * 2048);
    };

    char *const bias_slot = smem + BIAS_OFF + wid * 1024;
    const bool lds_bias = p.bias != nullptr;
    // 512 bytes per wave: lanes 0..31 carry 16 bytes each, the others point past the end of the vector (zeros into
    // the unused half of the wave's slot)
    const __amdgpu_buffer_rsrc_t rsrc_bias = bounded_rsrc(p.bias, lds_bias ? (unsigned)p.N * 4u : 0u);
    const unsigned bias_voff = lane < 32 ? (unsigned)lane * 16u : 0xfffffff0u;

    // ---- prologue: k-tiles 0 and 1 (the host guarantees nk >= 2), fragments of k-tile 0's first k-step ----
    stage_all(0);
    cursor_next();
    stage_all(DBUF_BYTES);
    cursor_next();
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#define TC_W_FOR16(F)                                                                                                  \
    F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(8) F(9) F(10) F(11) F(12) F(13) F(14) F(15)
#define TC_W_READ0(i) read_frag(smem, 0, std::integral_constant<int, i>{});
    TC_W_FOR16(TC_W_READ0)
#undef TC_W_READ0

    f32x4 acc[8][8];                             // [column tile ni = 2 cb + nb][row block mi], D[n][m]
#ifdef TC_GEMM_STAMPS
    unsigned long long c_st[6] = {};
    TC_WSTAMP(k_begin);
#endif
    int g = 0;
    int tm = tm0, tn = tn0, ti = tile0;
    bool stores_pending = false;                 // the previous tile's epilogue issued all of its 32 stores
    for (int t = 0; t < my_count; ++t) {
#pragma unroll
        for (int ni = 0; ni < 8; ++ni)
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < p.nk; ++kt, ++g) {
            const int cur = (g & 1) * DBUF_BYTES, nxt = DBUF_BYTES - cur;
            const char *sb = smem + cur, *sn = smem + nxt;
            // ---- first half: k-step 0; between the MFMAs the fragments of k-step 1 ----
            TC_WSTAMP(s0);
            // Sixteen groups of four MFMAs.  The sixteen fragment reads of a half go out behind the first twelve
            // groups (two behind each of the first four), so that the last one has four groups to land before
            // the next half begins; the pieces go out one behind each group.
            __builtin_amdgcn_sched_barrier(0);
#define TC_W_GROUP(KS, i)                                                                                              \
                {                                                                                                      \
                    constexpr int ni = (i) >> 1, m0 = ((i) & 1) * 4;                                                   \
                    _Pragma("unroll") for (int mi = m0; mi < m0 + 4; ++mi)                                             \
                        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[KS][ni]),  \
                                                                              __builtin_bit_cast(bf16x8, fa[KS][mi]),  \
                                                                              acc[ni][mi], 0, 0, 0);                   \
                }
#define TC_W_READS(buf, KS, i)                                                                                         \
                if constexpr ((i) < 4) {                                                                               \
                    read_frag(buf, KS, std::integral_constant<int, 2 * (i)>{});                                        \
                    read_frag(buf, KS, std::integral_constant<int, 2 * (i) + 1>{});                                    \
                } else if constexpr ((i) < 12) {                                                                       \
                    read_frag(buf, KS, std::integral_constant<int, (i) + 4>{});                                        \
                }
#define TC_W_HALF_A(i)                                                                                                 \
            {                                                                                                          \
                TC_W_GROUP(0, i)                                                                                       \
                TC_W_READS(sb, 1, i)                                                                                   \
                __builtin_amdgcn_sched_barrier(0);                                                                     \
            }
            TC_W_FOR16(TC_W_HALF_A)
#undef TC_W_HALF_A
            TC_WSTAMP(s1);
            // ---- the k-tile's one wait and barrier ----
            // In flight: the 16 pieces of k-tile g + 1 (+ the bias piece) and, in a tile's first k-tile, the previous
            // tile's 32 stores behind them (the counter retires in issue order).
            if (kt == 0 && stores_pending) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            TC_WSTAMP(s2);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            TC_WSTAMP(s3);
            // this tile's epilogue inputs ride with the pieces of its second k-tile: the bias of the wave's 128 columns
            if (kt == 1 && lds_bias)
                tc_blds16(rsrc_bias, bias_voff, (unsigned)(tn * BN + wc * 128) * 4u, bias_slot);
            // ---- second half: k-step 1; between the MFMAs the fragments of k-tile g + 1, k-step 0, and the
            // pieces of k-tile g + 2 into the buffer that every wave has finished reading ----
            // (after the cursor has run past the last tile it stays on that tile's last k-tile: the pieces are then
            // issued again, into a buffer nobody reads any more -- one code path, or the compiler copies all the
            // accumulators where two paths meet)
#define TC_W_HALF_B(i)                                                                                                 \
            {                                                                                                          \
                TC_W_GROUP(1, i)                                                                                       \
                TC_W_READS(sn, 0, i)                                                                                   \
                stage_piece(cur, std::integral_constant<int, i>{});                                                    \
                __builtin_amdgcn_sched_barrier(0);                                                                     \
            }
            TC_W_FOR16(TC_W_HALF_B)
#undef TC_W_HALF_B
            cursor_next();
            TC_WSTAMP(s4);
            TC_WADD(0, s0, s1); TC_WADD(1, s1, s2); TC_WADD(2, s2, s3); TC_WADD(3, s3, s4);
#undef TC_W_GROUP
#undef TC_W_READS
        }

        TC_WSTAMP(e0);
        // ---- epilogue: bias, bf16, lane exchange (four neighbouring lanes write 64 contiguous bytes of a row) ----
        {
            int L = __builtin_amdgcn_mbcnt_hi(-1, __builtin_amdgcn_mbcnt_lo(-1, 0));
            asm volatile("" : "+v"(L));                 // made here: nothing of the epilogue lives through the k-loop
            const int pull = (((L & 3) << 4) | (L >> 2)) << 2;
            const int trow = L >> 2, tseg = L & 3;
            const int mbase = tm * BM + wr * 128 + trow, nbase = tn * BN + wc * 128 + tseg * 8;
            const bool full = tm * BM + BM <= p.M;
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
                if (lds_bias) {
                    b0 = *reinterpret_cast<const f32x4 *>(bias_slot + (cb * 32 + (L >> 4) * 8) * 4);
                    b1 = *reinterpret_cast<const f32x4 *>(bias_slot + (cb * 32 + (L >> 4) * 8 + 4) * 4);
                }
#pragma unroll
                for (int mi = 0; mi < 8; ++mi) {
                    const f32x4 v = acc[2 * cb][mi] + b0, v2 = acc[2 * cb + 1][mi] + b1;
                    u32x4 w;
                    w[0] = tc_pack_bf16(v[0], v[1]);
                    w[1] = tc_pack_bf16(v[2], v[3]);
                    w[2] = tc_pack_bf16(v2[0], v2[1]);
                    w[3] = tc_pack_bf16(v2[2], v2[3]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[e] = (unsigned)__builtin_amdgcn_ds_bpermute(pull, (int)w[e]);
                    const int m = mbase + mi * 16;
                    if (m < p.M) *reinterpret_cast<u32x4 *>(p.out + ((long)m * p.ldo + nbase + cb * 32) * 2) = w;
                    // four accumulator pairs at a time: the scheduler would otherwise copy a whole tile of
                    // accumulation registers into vector registers at once, and spill around the k-loop for it
                    if ((mi & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
            }
            stores_pending = full;
        }
        TC_WSTAMP(e1);
        TC_WADD(4, e0, e1);
        advance(tm, tn, ti);
    }
#undef TC_W_FOR16
#ifdef TC_GEMM_STAMPS
    if (p.stamps && threadIdx.x == 0) {
        TC_WSTAMP(k_end);
        unsigned long long *o = p.stamps + blockIdx.x * 8;
        for (int i = 0; i < 5; ++i) o[i] = c_st[i];
        o[5] = k_end - k_begin; o[6] = (unsigned long long)my_count * p.nk; o[7] = my_count;
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // no piece may land after the workgroup has given up its LDS
}

}  // namespace

int tc_gemm256w_supported(int epilogue, const TcGemmParams &p) {
    static int on = -1;
    if (on < 0) { const char *e = getenv("TC_GEMM_W"); on = e ? atoi(e) : 0; }
    if (!on) return 0;
    if (epilogue != TC_EPI_BIAS) return 0;
    if (p.nk < 2 || p.panels || p.ln_rows || p.row_acc) return 0;
    // 32-bit byte offsets, rows of the last tile past M included (they read as zero through the bounded descriptor)
    if (((long)p.M + 256) * p.lda_bytes >= (1L << 32) || (long)p.N * p.ldw_bytes >= (1L << 32)) return 0;
    return 1;
}

int tc_gemm256w_launch(int epilogue, const TcGemmParams &p, int grid, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(gemm256w_kernel<TC_EPI_BIAS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kw::LDS_BYTES);
        if (e != hipSuccess) {
            tc_set_error("tc_gemm: hipFuncSetAttribute: %s", hipGetErrorString(e));
            return TC_ELAUNCH;
        }
        attr_set = true;
    }
    TC_LAUNCH((gemm256w_kernel<TC_EPI_BIAS>), dim3(grid), dim3(256), kw::LDS_BYTES, stream, p);
    TC_CHECK_LAUNCH("tc_gemm");
    return TC_OK;
}
