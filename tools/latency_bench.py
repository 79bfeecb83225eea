#!/usr/bin/env python3
"""Latency of small inference batches, eager against the HIP-graph replay (ProteinBertModel.capture_forward):
forward + heads of B x T tokens, bf16 and f32, median of 50 calls with a synchronisation after each."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import transcodon_amd as ta  # noqa: E402

dev = "cuda"
model = ta.ProteinBertModel(ta.default_args(), ta.Alphabet.from_architecture("CodonModel"))
ta.init_weights(model, seed=1)
model = model.to(dev).eval()
model.check_indices = False


def median_ms(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts))


for dtype in (torch.bfloat16, torch.float32):
    model.set_compute_dtype(dtype)
    for B, T in ((1, 131), (1, 512), (1, 1666), (8, 512), (32, 512)):
        rng = np.random.default_rng(B * 1000 + T)
        tok = torch.from_numpy(rng.choice([5, 6, 7, 8, 9], (B, T))).to(dev)
        tok[:, 0], tok[:, -1] = 0, 2
        sp = torch.from_numpy(rng.integers(0, 1438, B)).to(dev)
        with torch.no_grad():
            eager = median_ms(lambda: model(tok, sp))
        graphed = model.capture_forward(B, T)
        replay = median_ms(lambda: graphed(tok, sp))
        print(f"{str(dtype)[6:]:9s} {B:3d} x {T:4d}: eager {eager:7.3f} ms   graph {replay:7.3f} ms   x{eager / replay:.2f}", flush=True)
        del graphed
