#!/usr/bin/env python3
"""End-to-end rate of the batched codon-optimisation driver (optimize_codons: host tokenisation, packed
forward, decode, back-translation check, strings out) on 2048 proteins of 200-480 amino acids, next to
the string tokenisation path of the reference's row loop (infer.py:261-286) on the host alone."""
import time, numpy as np, torch, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import transcodon_amd as ta
from transcodon_amd import infer_utr as I
rng = np.random.default_rng(0)
AAS = "ACDEFGHIKLMNPQRSTVWY"
n = 2048
prots = ["M" + "".join(rng.choice(list(AAS), int(rng.integers(200, 480)))) for _ in range(n)]
orgs = [int(x) for x in rng.integers(0, 1438, n)]
t0 = time.perf_counter()
items = []
for i, prot in enumerate(prots):
    aa = prot + "_"
    tmpl = I.AA_tokenize(aa)
    seq = I.CodonSequence_infer(tmpl)
    ids = I._ALPHABET.encode(seq.seq)
    items.append(ids)
t1 = time.perf_counter()
ntok = sum(len(x) for x in items)
print(f"host tokenisation: {t1 - t0:.3f} s for {n} proteins, {ntok} tokens -> {ntok / (t1 - t0) / 1e6:.2f} M tokens/s")
if torch.cuda.is_available():
    model = ta.ProteinBertModel(ta.default_args(), ta.Alphabet.from_architecture("CodonModel"))
    ta.init_weights(model, seed=1)
    model = model.to("cuda").eval().set_compute_dtype(torch.bfloat16)
    for _ in range(2):
        t0 = time.perf_counter(); out = I.optimize_codons(model, prots, orgs); torch.cuda.synchronize(); t1 = time.perf_counter()
        print(f"optimize_codons end to end: {t1 - t0:.3f} s -> {ntok / (t1 - t0) / 1e6:.2f} M tokens/s")
