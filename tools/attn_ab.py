#!/usr/bin/env python3
"""A/B timing of tc_attn_fwd in two builds of the library, one process, interleaved rounds.

    python tools/attn_ab.py [base.so] [new.so] [--shapes 256x512,256x1024]
"""
import argparse
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transcodon_amd import _lib  # noqa: E402


def load(path, q2=False):
    """-> the library's forward entry: tc_attn_fwd, or tc_attn_fwd_q2 (q pre-multiplied by log2(e))"""
    lib = C.CDLL(path)
    res, args = _lib._SIGNATURES["tc_attn_fwd"]
    fn = lib.tc_attn_fwd_q2 if q2 else lib.tc_attn_fwd
    fn.restype, fn.argtypes = res, args
    return fn


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("base", nargs="?", default=os.path.join(ROOT, "tools", "ab", "libtc_base.so"))
    ap.add_argument("new", nargs="?", default=os.path.join(ROOT, "transcodon_amd", "csrc", "libtranscodon_hip.so"))
    ap.add_argument("--shapes", default="256x512,128x1024,32x2048")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--q2", action="store_true", help="time the new library's base-2 entry (tc_attn_fwd_q2)")
    ap.add_argument("--q2-base", action="store_true", help="... and the base library's too")
    a = ap.parse_args()
    q2 = {"base": a.q2_base, "new": a.q2 or a.q2_base}
    libs = {"base": load(a.base, q2["base"]), "new": load(a.new, q2["new"])}
    H, E = 12, 768
    strm = torch.cuda.current_stream().cuda_stream
    for shp in a.shapes.split(","):
        B, T = (int(v) for v in shp.split("x"))
        g = torch.Generator(device="cuda").manual_seed(1)
        qkv = torch.randn(B * T, 3 * E, device="cuda", generator=g)
        qkv[:, :E] *= 0.125
        qkv2 = qkv.clone()
        qkv2[:, :E] *= 1.4426950408889634
        qkv, qkv2 = qkv.bfloat16(), qkv2.bfloat16()
        outs = {k: torch.empty(B * T, E, device="cuda", dtype=torch.bfloat16) for k in libs}

        def run(k):
            src = qkv2 if q2[k] else qkv
            assert libs[k](1, src.data_ptr(), 3 * E, None, outs[k].data_ptr(), B, T, H, 64, strm) == 0

        for k in libs:
            run(k)
        torch.cuda.synchronize()
        maxd = float((outs["base"].float() - outs["new"].float()).abs().max())
        times = {k: [] for k in libs}
        for _ in range(a.rounds):
            for k in libs:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    run(k)
                e1.record()
                torch.cuda.synchronize()
                times[k].append(e0.elapsed_time(e1) / a.iters * 1e3)
        med = {k: sorted(v)[len(v) // 2] for k, v in times.items()}
        fl = 4.0 * B * H * T * T * 64
        print(f"{B}x{T}: base {med['base']:7.1f} us  new {med['new']:7.1f} us  new/base {med['new'] / med['base']:.3f}  "
              f"{fl / med['new'] / 1e6:7.1f} TF/s  max|d|={maxd:.3g}", flush=True)


if __name__ == "__main__":
    main()
