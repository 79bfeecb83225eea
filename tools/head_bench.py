#!/usr/bin/env python3
"""Times tc_head_logits_fwd at the inference shape (131072 rows of a [M, 1536] bf16 buffer, V = 10 and 3)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transcodon_amd import _lib, ops  # noqa: E402

dev = "cuda"
torch.manual_seed(3)
M, E = 131072, 768
g = torch.randn(M, 2 * E, device=dev).bfloat16()
gam, bet = torch.randn(E, device=dev), torch.randn(E, device=dev)
lib = _lib.load()
s = torch.cuda.current_stream().cuda_stream
for V, col0 in ((10, 0), (3, E)):
    W = torch.randn(V, E, device=dev)
    b = torch.randn(V, device=dev)
    out = torch.empty(M, V, device=dev)

    def run():
        assert lib.tc_head_logits_fwd(1, g.data_ptr(), 2 * E, col0, gam.data_ptr(), bet.data_ptr(), W.data_ptr(),
                                      b.data_ptr(), out.data_ptr(), M, E, V, 1e-5, s) == 0
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"V={V:2d}: {us:7.1f} us  {M * E * 2 / us / 1e6:5.2f} TB/s  checksum {out.double().sum().item():.6e}", flush=True)
