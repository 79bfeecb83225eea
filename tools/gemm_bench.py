#!/usr/bin/env python3
"""Micro-benchmark of tc_gemm on the four encoder shapes (M = 131072 tokens), random data.

    python tools/gemm_bench.py [--iters 20] [--only qkv,out,fc1,fc2]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transcodon_amd import _lib, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--M", type=int, default=131072)
    ap.add_argument("--only", default="qkv,out,fc1,fc2")
    ap.add_argument("--dtype", default="bf16")
    args = ap.parse_args()
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    dev = "cuda"
    M, E, F, T = args.M, 768, 3072, 512
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(M, E, device=dev, generator=g).to(dt)
    mid = torch.randn(M, F, device=dev, generator=g).to(dt)
    res = torch.randn(M, E, device=dev, generator=g)
    ang = torch.arange(T, dtype=torch.float32)[:, None] * (1.0 / 10000 ** (torch.arange(0, 64, 2).float() / 64))[None]
    cos, sin = ang.cos().to(dev).contiguous(), ang.sin().to(dev).contiguous()
    shapes = {
        "qkv": (x, 3 * E, E, _lib.EPI_QKV_ROPE),
        "out": (x, E, E, _lib.EPI_BIAS_RESIDUAL),
        "fc1": (x, F, E, _lib.EPI_BIAS_GELU),
        "fc2": (mid, E, F, _lib.EPI_BIAS_RESIDUAL),
    }
    for name in args.only.split(","):
        a, N, K, epi = shapes[name]
        w = (torch.randn(N, K, device=dev, generator=g) * 0.03).to(dt)
        b = torch.randn(N, device=dev, generator=g) * 0.1
        out = torch.empty(M, N, device=dev, dtype=torch.float32 if epi == _lib.EPI_BIAS_RESIDUAL else dt)
        kw = dict(residual=res, out=out) if epi == _lib.EPI_BIAS_RESIDUAL else dict(out=out)
        if epi == _lib.EPI_QKV_ROPE:
            kw.update(q_scale=0.125, rope_cos=cos, rope_sin=sin, T=T, head_dim=64)
        for _ in range(3):
            ops.gemm(a, w, b, epi, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            ops.gemm(a, w, b, epi, **kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        print(f"{name:4s} M={M} N={N} K={K}: {ms * 1e3:8.1f} us  {2.0 * M * N * K / ms / 1e9:8.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
