"""Epilogue cost of the large-tile tc_gemm: bias-only against bias + GELU, short and long K."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transcodon_amd import _lib, ops  # noqa: E402

dev = "cuda"
M = 131072
for (N, K) in [(3072, 768), (3072, 3072), (768, 3072)]:
    a = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.03).bfloat16()
    b = torch.randn(N, device=dev) * 0.1
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for name, epi in (("bias", _lib.EPI_BIAS), ("gelu", _lib.EPI_BIAS_GELU)):
        for _ in range(3):
            ops.gemm(a, w, b, epi, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.gemm(a, w, b, epi, out=out)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"N={N} K={K} {name}: {us:8.1f} us {2 * M * N * K / us / 1e6:8.1f} TF/s", flush=True)
