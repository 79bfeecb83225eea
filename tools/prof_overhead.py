#!/usr/bin/env python3
"""Step time of the forward + decode with and without the in-library launch timers.

    python tools/prof_overhead.py [B T]        (default 256 512, the bench workload)
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import transcodon_amd as ta  # noqa: E402
from transcodon_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda", 0)
B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (bench.B_PER_GPU, bench.SEQ_LEN)
model = ta.ProteinBertModel(ta.default_args(), ta.Alphabet.from_architecture("CodonModel"))
ta.init_weights(model, seed=1234)
model = model.to(dev).eval().set_compute_dtype(torch.bfloat16)
model.check_indices = False
tok_np, sp_np, codes, offs, row0 = bench.synthetic_batch(B, T, seed=1234)
tok, sp = torch.from_numpy(tok_np).to(dev), torch.from_numpy(sp_np).to(dev)
aa = torch.frombuffer(bytearray(codes), dtype=torch.uint8).to(dev)
offs_d, row0_d = torch.from_numpy(offs).to(dev), torch.from_numpy(row0).to(dev)
lib = _lib.load()


def step():
    with torch.no_grad():
        out = model(tok, sp)
    return ops.codon_decode(out["logits"].view(B * T, -1), aa, offs_d, row0_d)


for _ in range(3):
    step()
for rnd in range(3):
    for on in (0, 1):
        lib.tc_prof_reset()
        lib.tc_prof_enable(on)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 100
        lib.tc_prof_enable(0)
        print(f"{B}x{T} launch timers {'on ' if on else 'off'}: {ms:.3f} ms/step  {B * T / ms / 1e3:.3f} M tokens/s", flush=True)
