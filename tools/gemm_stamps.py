#!/usr/bin/env python3
"""Cycle breakdown of the persistent large-tile GEMM (debug build tools/libtc_stamps.so,
compiled with -DTC_GEMM_STAMPS): wave 0 of every workgroup accumulates s_memtime deltas for
wait (vmcnt+barrier), load issue, math (LDS reads + MFMA) and epilogue."""
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
stamps = torch.zeros(256 * 4, dtype=torch.int64, device=dev)
lib.tc_debug_set_stamp_buffer(stamps.data_ptr())
for name, a, N, K, epi in (("qkv", x, 3 * E, E, 3), ("out", x, E, E, 2), ("fc1", x, F, E, 1), ("fc2", mid, E, F, 2)):
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
    st = stamps.view(256, 4).double().cpu()
    tiles = ((M + 255) // 256) * (N // 256) / 256.0
    mean = st.mean(0)
    tot = float(mean.sum())
    print(f"{name}: tiles/WG {tiles:.1f} k-tiles/tile {K // 64}  cycles/WG total {tot:.0f}  wait {mean[0]:.0f} ({mean[0] / tot:.2f}) "
          f"issue {mean[1]:.0f} ({mean[1] / tot:.2f}) math {mean[2]:.0f} ({mean[2] / tot:.2f}) epilogue {mean[3]:.0f} ({mean[3] / tot:.2f})  "
          f"per k-tile: wait {mean[0] / tiles / (K // 64):.0f} math {mean[2] / tiles / (K // 64):.0f}; per tile epilogue {mean[3] / tiles:.0f}", flush=True)
