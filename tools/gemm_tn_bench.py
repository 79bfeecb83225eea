#!/usr/bin/env python3
"""Micro-benchmark of tc_gemm_tn (weight gradients dW = dY^T X) on the encoder's four shapes,
M = 65536 tokens, splits as the training path picks them (256 / tiles)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transcodon_amd import ops  # noqa: E402

dev = "cuda"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for name, I, J in (("fc2", 768, 3072), ("fc1", 3072, 768), ("qkv", 2304, 768), ("out", 768, 768)):
    a = torch.randn(M, I, device=dev).bfloat16()
    b = torch.randn(M, J, device=dev).bfloat16()
    splits = max(1, 256 // ((I // 256) * (J // 256)))
    for _ in range(3):
        ops.gemm_tn(a, b, splits)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.gemm_tn(a, b, splits)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{name:4s} I={I} J={J} M={M} splits={splits}: {us:8.1f} us {2.0 * M * I * J / us / 1e6:8.1f} TFLOP/s", flush=True)
