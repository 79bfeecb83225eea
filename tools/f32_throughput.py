import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import transcodon_amd as ta
from transcodon_amd import _lib
model = ta.ProteinBertModel(ta.default_args(), ta.Alphabet.from_architecture("CodonModel"))
ta.init_weights(model, seed=1)
model = model.to("cuda").eval().set_compute_dtype(torch.float32)
model.check_indices = False
lib = _lib.load()
for B, T in ((64, 512), (8, 512)):
    rng = np.random.default_rng(0)
    tok = torch.from_numpy(rng.choice([5, 6, 7, 8, 9], (B, T))).cuda(); tok[:, 0] = 0; tok[:, -1] = 2
    sp = torch.from_numpy(rng.integers(0, 1438, B)).cuda()
    with torch.no_grad():
        for _ in range(2): model(tok, sp)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): model(tok, sp)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        lib.tc_prof_reset(); lib.tc_prof_enable(1)
        model(tok, sp); torch.cuda.synchronize()
        lib.tc_prof_enable(0)
    prof = {n: _lib.prof_collect(k) for n, k in (("gemm", _lib.PROF_GEMM), ("attn", _lib.PROF_ATTN), ("ln", _lib.PROF_LAYERNORM))}
    print(f"f32 {B}x{T}: {dt*1e3:.1f} ms = {B*T/dt/1e3:.0f} k tokens/s", {k: v for k, v in prof.items()})
