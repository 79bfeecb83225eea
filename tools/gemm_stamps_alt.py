#!/usr/bin/env python3
"""Per-k-tile cycle profile of the alternating-group large-tile GEMM (debug build
tools/libtc_stamps.so, -DTC_GEMM_STAMPS; run with TC_GEMM_ALT=1): waves 0 and 4 of every
workgroup accumulate s_memtime deltas per k-tile position inside a tile, and for the epilogue."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get("TC_STAMPS_LIB", "libtc_stamps.so")))
vp = C.c_void_p
lib.tc_gemm.restype = C.c_int
lib.tc_gemm.argtypes = [C.c_int, C.c_int, vp, C.c_int, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int,
                        C.c_float, vp, vp, C.c_int, C.c_int, vp]
lib.tc_debug_set_stamp_buffer.argtypes = [vp]
lib.tc_last_error.restype = C.c_char_p
lib.tc_gemm_train.restype = C.c_int
lib.tc_gemm_train.argtypes = [C.c_int, C.c_int, vp, C.c_int, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]
lib.tc_gemm_residual_split.restype = C.c_int
lib.tc_gemm_residual_split.argtypes = [C.c_int, vp, C.c_int, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]

dev = "cuda"
M, E, F, T = 131072, 768, 3072, 512
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(M, E, device=dev, generator=g).bfloat16()
mid = torch.randn(M, F, device=dev, generator=g).bfloat16()
res = torch.randn(M, E, device=dev, generator=g)
ang = torch.arange(T, dtype=torch.float32)[:, None] * (1.0 / 10000 ** (torch.arange(0, 64, 2).float() / 64))[None]
cos, sin = ang.cos().to(dev).contiguous(), ang.sin().to(dev).contiguous()
stamps = torch.zeros(256 * 2 * 32 + 256 * 2 * 26, dtype=torch.int64, device=dev)
lib.tc_debug_set_stamp_buffer(stamps.data_ptr())
SHAPES = {"bias": (x, F, E, 0), "fc1": (x, F, E, 1), "qkv": (x, 3 * E, E, 3), "out": (x, E, E, 2),
          "out_f32": (x, E, E, 4), "out_bf16": (x, E, E, 0),
          "fc2": (mid, E, F, 2), "fc2_f32": (mid, E, F, 4), "fc2_bf16": (mid, E, F, 0),
          # residual stream as two bf16 planes updated in place (TC_RES_STREAM=0: accumulators preloaded)
          "out_split": (x, E, E, 5), "fc2_split": (mid, E, F, 5)}
plane_hi = torch.randn(M, E, device=dev, generator=g).bfloat16()
plane_lo = (torch.randn(M, E, device=dev, generator=g) * 1e-3).bfloat16()
row_acc = torch.zeros(M, 2, device=dev, dtype=torch.int64)
names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["bias", "fc1", "qkv", "out"]
for name in names:
    a, N, K, epi = SHAPES[name]
    w = (torch.randn(N, K, device=dev, generator=g) * 0.03).bfloat16()
    b = torch.randn(N, device=dev, generator=g) * 0.1
    ldo = N + 64 if N == F else N      # rows of 6144 bytes store slowly: pad them
    out = torch.empty(M, ldo, device=dev, dtype=torch.float32 if epi in (2, 4, 5) else torch.bfloat16)
    s = torch.cuda.current_stream().cuda_stream
    for it in range(int(os.environ.get("TC_STAMP_REPS", "3"))):
        stamps.zero_()
        if epi == 5:
            rc = lib.tc_gemm_residual_split(1, a.data_ptr(), a.shape[1], w.data_ptr(), b.data_ptr(), plane_hi.data_ptr(), plane_lo.data_ptr(),
                                            row_acc.data_ptr(), ldo, M, N, K, s)
        elif epi == 4:
            rc = lib.tc_gemm_train(1, 4, a.data_ptr(), a.shape[1], w.data_ptr(), None, None, out.data_ptr(), None, ldo, M, N, K, 1, s)
        else:
            rc = lib.tc_gemm(1, epi, a.data_ptr(), a.shape[1], w.data_ptr(), None if epi == 4 else b.data_ptr(), res.data_ptr() if epi == 2 else None,
                             out.data_ptr(), ldo, M, N, K, 0.125, cos.data_ptr(), sin.data_ptr(), T, 64, s)
        assert rc == 0, lib.tc_last_error()
    torch.cuda.synchronize()
    st = stamps[:256 * 2 * 32].view(256, 2, 32).double().cpu()
    ph = stamps[256 * 2 * 32:].view(256, 2, 2, 13).double().cpu()
    for grp in (0, 1):
        m = st[:, grp].mean(0)
        tiles = float(m[26])
        per = [round(float(v) / tiles) for v in m[:24] if v > 0]
        ghz = float(m[25]) / max(float(m[27]), 1.0) * 0.1      # shader clocks per 100 MHz tick
        print(f"{name} group {grp}: in-kernel clock {ghz:.2f} GHz  tiles/WG {tiles:.1f} total {m[25]:.0f} per tile {m[25] / tiles:.0f}  epilogue/tile {m[24] / tiles:.0f}  "
              f"k-tile by position {per}", flush=True)
        if float(ph.sum()) > 0:          # library built with -DTC_GEMM_PHASES: a tile's first and last k-tile
            for c, cname in enumerate(os.environ.get("TC_PH_NAMES", "first,last").split(",")):
                v = (ph[:, grp, c].mean(0) / tiles).tolist()
                print(f"    {cname:5s} k-tile: issue+epilogue {v[0]:.0f} | reads {v[1]:.0f} | bar {v[2]:.0f} | mfma+bar {v[3]:.0f} || "
                      f"p1: {v[4]:.0f} {v[5]:.0f} {v[6]:.0f} | p2: {v[7]:.0f} {v[8]:.0f} {v[9]:.0f} | p3: {v[10]:.0f} {v[11]:.0f} {v[12]:.0f}", flush=True)
