#!/usr/bin/env python3
"""One f32 sequence through the forward a few times (for `rocprofv3 --kernel-trace --stats`): which kernels make up
the latency of the reference's own single-sequence use."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import transcodon_amd as ta  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 131
model = ta.ProteinBertModel(ta.default_args(), ta.Alphabet.from_architecture("CodonModel"))
ta.init_weights(model, seed=1)
model = model.to("cuda").eval().set_compute_dtype(torch.float32)
rng = np.random.default_rng(0)
tok = torch.from_numpy(rng.choice([5, 6, 7, 8, 9], (1, T))).cuda()
sp = torch.tensor([3]).cuda()
with torch.no_grad():
    for _ in range(20):
        model(tok, sp)
torch.cuda.synchronize()
