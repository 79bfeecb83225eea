#!/usr/bin/env python3
"""Per-k-tile cycle profile of the alternating-group large-tile GEMM (debug build
tools/libtc_stamps.so, -DTC_GEMM_STAMPS; run with TC_GEMM_ALT=1): waves 0 and 4 of every
workgroup accumulate s_memtime deltas per k-tile position inside a tile, and for the epilogue."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libtc_stamps.so"))
vp = C.c_void_p
lib.tc_gemm.restype = C.c_int
lib.tc_gemm.argtypes = [C.c_int, C.c_int, vp, C.c_int, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int,
                        C.c_float, vp, vp, C.c_int, C.c_int, vp]
lib.tc_debug_set_stamp_buffer.argtypes = [vp]

dev = "cuda"
M, E, F, T = 131072, 768, 3072, 512
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(M, E, device=dev, generator=g).bfloat16()
mid = torch.randn(M, F, device=dev, generator=g).bfloat16()
res = torch.randn(M, E, device=dev, generator=g)
ang = torch.arange(T, dtype=torch.float32)[:, None] * (1.0 / 10000 ** (torch.arange(0, 64, 2).float() / 64))[None]
cos, sin = ang.cos().to(dev).contiguous(), ang.sin().to(dev).contiguous()
stamps = torch.zeros(256 * 2 * 32, dtype=torch.int64, device=dev)
lib.tc_debug_set_stamp_buffer(stamps.data_ptr())
for name, a, N, K, epi in (("fc1", x, F, E, 1), ("qkv", x, 3 * E, E, 3), ("fc2", mid, E, F, 2)):
    w = (torch.randn(N, K, device=dev, generator=g) * 0.03).bfloat16()
    b = torch.randn(N, device=dev, generator=g) * 0.1
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if epi == 2 else torch.bfloat16)
    s = torch.cuda.current_stream().cuda_stream
    for it in range(3):
        stamps.zero_()
        rc = lib.tc_gemm(1, epi, a.data_ptr(), K, w.data_ptr(), b.data_ptr(), res.data_ptr() if epi == 2 else None,
                         out.data_ptr(), N, M, N, K, 0.125, cos.data_ptr(), sin.data_ptr(), T, 64, s)
        assert rc == 0
    torch.cuda.synchronize()
    st = stamps.view(256, 2, 32).double().cpu()
    for grp in (0, 1):
        m = st[:, grp].mean(0)
        tiles = float(m[26])
        per = [round(float(v) / tiles) for v in m[:24] if v > 0]
        print(f"{name} group {grp}: tiles/WG {tiles:.1f} total {m[25]:.0f} per tile {m[25] / tiles:.0f}  epilogue/tile {m[24] / tiles:.0f}  "
              f"k-tile by position {per}", flush=True)
