#!/usr/bin/env python3
"""Times tc_layernorm_fwd at the inference shape (131072 x 768, f32 in, bf16 out)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transcodon_amd import ops  # noqa: E402

dev = "cuda"
M, E = 131072, 768
x = torch.randn(M, E, device=dev)
g, b = torch.randn(E, device=dev), torch.randn(E, device=dev)
for _ in range(3):
    ops.layernorm(x, g, b, out_dtype=torch.bfloat16)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.layernorm(x, g, b, out_dtype=torch.bfloat16)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print(f"layernorm fwd {us:7.1f} us  {M * E * 6 / us / 1e6:6.2f} TB/s", flush=True)
