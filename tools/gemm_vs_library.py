#!/usr/bin/env python3
"""Calibration only (not a product path): the encoder's four GEMM shapes through torch.matmul (hipBLASLt / rocBLAS,
bf16 in, bf16 out, no epilogue) beside tc_gemm with the bias epilogue, same process, same data."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transcodon_amd import _lib, ops  # noqa: E402

M = 131072
dev = "cuda"


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for name, N, K in (("qkv", 2304, 768), ("out", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072), ("sq", 3072, 3072)):
    g = torch.Generator(device=dev).manual_seed(1)
    a = torch.randn(M, K, device=dev, generator=g).bfloat16()
    w = (torch.randn(N, K, device=dev, generator=g) * 0.03).bfloat16()
    b = torch.zeros(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t_lib = timeit(lambda: torch.matmul(a, w.t(), out=out))
    t_tc = timeit(lambda: ops.gemm(a, w, b, _lib.EPI_BIAS, out=out))
    fl = 2.0 * M * N * K
    print(f"{name:4s} N={N:5d} K={K:5d}: library {t_lib:7.1f} us {fl / t_lib / 1e6:7.1f} TF/s   tc_gemm {t_tc:7.1f} us "
          f"{fl / t_tc / 1e6:7.1f} TF/s   tc/library {t_tc / t_lib:.3f}", flush=True)
