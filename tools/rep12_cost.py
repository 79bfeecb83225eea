#!/usr/bin/env python3
"""What representations[12] costs a large folded bf16 forward (256 x 512 tokens, random-init full model): with it the
final LayerNorm runs as a kernel and writes its f32 output; without it emb_layer_norm_after is part of the head dense
products.  Prints ms per forward (HIP graph replays, one stream) for both."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import transcodon_amd as ta  # noqa: E402

torch.manual_seed(0)
model = ta.ProteinBertModel(ta.default_args(), ta.Alphabet.from_architecture("CodonModel")).to("cuda").eval()
model.set_compute_dtype(torch.bfloat16)
B, T = 256, 512
tok = torch.randint(5, 69, (B, T), device="cuda", dtype=torch.int32)
sp = torch.randint(0, 1000, (B,), device="cuda", dtype=torch.int32)
for reps in ((12,), ()):
    g = model.capture_forward(B, T, repr_layers=reps)
    for _ in range(3):
        g(tok, sp, check_indices=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        g(tok, sp, check_indices=False)
    torch.cuda.synchronize()
    print(f"repr_layers={list(reps)}: {1e3 * (time.perf_counter() - t0) / n:.3f} ms per forward", flush=True)
