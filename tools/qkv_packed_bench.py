#!/usr/bin/env python3
"""q/k/v product timing: padded form (alternating large-tile kernel, rotary factors from LDS table slices)
against the packed form (a rotary position per row: single-group kernel, per-row gather), with all tiles
inside one sequence (positions 0..511 repeating) and with a sequence boundary in every fourth tile.
TC_GEMM_ALT=0 in the environment runs the padded form on the single-group kernel as well."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transcodon_amd import _lib  # noqa: E402

lib = _lib.load()
dev = "cuda"
M, E, T = 131072, 768, 2048
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(M, E, device=dev, generator=g).bfloat16()
w = (torch.randn(3 * E, E, device=dev, generator=g) * 0.03).bfloat16()
b = torch.randn(3 * E, device=dev, generator=g) * 0.1
ang = torch.arange(T, dtype=torch.float32)[:, None] * (1.0 / 10000 ** (torch.arange(0, 64, 2).float() / 64))[None]
cos, sin = ang.cos().to(dev).contiguous(), ang.sin().to(dev).contiguous()
out = torch.empty(M, 3 * E, device=dev, dtype=torch.bfloat16)
s = torch.cuda.current_stream().cuda_stream
pos_a = (torch.arange(M, device=dev) % 512).int()
lens = torch.tensor([1000, 1048] * 64, device=dev)[: M // 1024]
pos_b = torch.cat([torch.arange(int(n), device=dev) for n in lens.tolist()]).int()[:M].contiguous()


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3


def padded():
    _lib.check(lib.tc_gemm(1, _lib.EPI_QKV_ROPE, x.data_ptr(), E, w.data_ptr(), b.data_ptr(), None, out.data_ptr(), 3 * E,
                           M, 3 * E, E, 0.125, cos.data_ptr(), sin.data_ptr(), 512, 64, s), "tc_gemm")


def packed(pos):
    _lib.check(lib.tc_gemm_qkv_packed(1, x.data_ptr(), E, w.data_ptr(), b.data_ptr(), out.data_ptr(), 3 * E, M, 3 * E, E,
                                      0.125, cos.data_ptr(), sin.data_ptr(), pos.data_ptr(), T, 64, s), "tc_gemm_qkv_packed")


print(f"padded {timed(padded):.1f} us   packed, no boundary {timed(lambda: packed(pos_a)):.1f} us   "
      f"packed, boundaries {timed(lambda: packed(pos_b)):.1f} us")
