#!/usr/bin/env python3
"""Times tc_layernorm_bwd_dy at the training shape (65536 x 768, accumulate) for f32 and bf16 dy."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transcodon_amd import ops  # noqa: E402

dev = "cuda"
M, E = 65536, 768
x = torch.randn(M, E, device=dev)
g = torch.randn(E, device=dev)
dx = torch.zeros(M, E, device=dev)
dg, db = torch.zeros(E, device=dev), torch.zeros(E, device=dev)
for dt in (torch.float32, torch.bfloat16):
    dy = torch.randn(M, E, device=dev).to(dt)
    for _ in range(3):
        ops.layernorm_bwd(x, g, dy, dg, db, dx=dx, accumulate=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.layernorm_bwd(x, g, dy, dg, db, dx=dx, accumulate=True)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    nbytes = M * E * (4 + dy.element_size() + 8)
    print(f"dy {str(dt):16s} {us:7.1f} us  {nbytes / us / 1e6:6.2f} TB/s", flush=True)
