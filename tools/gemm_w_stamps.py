#!/usr/bin/env python3
"""Where a k-tile of the one-wave-per-SIMD large-tile GEMM (gemm_w.hip) spends its clocks (debug build
tools/libtc_stamps.so, -DTC_GEMM_STAMPS; run with TC_GEMM_W=1): wave 0 of every workgroup adds up s_memtime
deltas of the first half, the wait, the barrier, the second half and the epilogue.
    TC_GEMM_W=1 python tools/gemm_w_stamps.py [qkv_bias,fc1_bias,sq]"""
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
dev = "cuda"
M, E, F = 131072, 768, 3072
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(M, E, device=dev, generator=g).bfloat16()
mid = torch.randn(M, F, device=dev, generator=g).bfloat16()
stamps = torch.zeros(256 * 2 * 32 + 256 * 2 * 26, dtype=torch.int64, device=dev)
lib.tc_debug_set_stamp_buffer(stamps.data_ptr())
SHAPES = {"qkv_bias": (x, 3 * E, E), "fc1_bias": (x, F, E), "out_bias": (x, E, E), "fc2_bias": (mid, E, F), "sq": (mid, F, F)}
for name in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["qkv_bias", "fc1_bias"]):
    a, N, K = SHAPES[name]
    w = (torch.randn(N, K, device=dev, generator=g) * 0.03).bfloat16()
    b = torch.randn(N, device=dev, generator=g) * 0.1
    ldo = N + 64 if N == F else N
    out = torch.empty(M, ldo, device=dev, dtype=torch.bfloat16)
    s = torch.cuda.current_stream().cuda_stream
    for it in range(3):
        stamps.zero_()
        rc = lib.tc_gemm(1, 0, a.data_ptr(), a.shape[1], w.data_ptr(), b.data_ptr(), None, out.data_ptr(), ldo, M, N, K, 0.125,
                         None, None, 0, 64, s)
        assert rc == 0, lib.tc_last_error()
    torch.cuda.synchronize()
    st = stamps[:256 * 8].view(256, 8).double().cpu()
    kt, tiles = st[:, 6], st[:, 7]
    per = lambda c: float((st[:, c] / kt).mean())
    print(f"{name}: k-tiles/WG {float(kt.mean()):.0f} tiles/WG {float(tiles.mean()):.1f} | per k-tile: first half {per(0):.0f}  wait {per(1):.0f}  barrier {per(2):.0f}  "
          f"second half {per(3):.0f} | epilogue per tile {float((st[:, 4] / tiles).mean()):.0f} | total per tile {float((st[:, 5] / tiles).mean()):.0f}", flush=True)
