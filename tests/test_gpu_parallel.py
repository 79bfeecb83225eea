"""Data-parallel gradient mean with two ranks on the GPU.  The box has one GPU and RCCL refuses two
ranks on one device, so the ranks share cuda:0 and talk through gloo (GradientSync then stages each
bucket through host memory); everything else -- the per-layer "gradients ready" callbacks of
tc_encoder_backward_notify, bucket ranges, no_sync accumulation, the division by the world size -- is
the code the RCCL path runs."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

import transcodon_amd as ta
from oracle import transcodon_ref as O
from oracle.weights import seeded_state_dict
from transcodon_amd import train as TR

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RANK_SCRIPT = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
import transcodon_amd as ta
from oracle import transcodon_ref as O
from oracle.weights import seeded_state_dict
from transcodon_amd import train as TR
from transcodon_amd.parallel import GradientSync
sys.path.insert(0, os.path.join({root!r}, "tests"))
from test_gpu_parallel import make_model, make_batch
rank = int(os.environ["RANK"])
dist.init_process_group("gloo")
m = make_model()
flat = TR.FlatParameters(m)
flat.grad_sync = GradientSync(flat)
for micro in range(2):                      # two accumulated micro-batches, exchange on the second only
    tok, sp, labels = make_batch(rank, micro)
    ctx = flat.grad_sync.no_sync() if micro == 0 else None
    if ctx: ctx.__enter__()
    out = m(tok, sp)
    TR.mlm_loss(out["logits"], labels).backward()
    if ctx: ctx.__exit__(None, None, None)
torch.cuda.synchronize()
torch.save(flat.grad.cpu(), os.path.join({out!r}, f"grad{{rank}}.pt"))
dist.barrier()
dist.destroy_process_group()
"""


def make_model():
    cfg = O.TINY
    args = ta.default_args(num_layers=cfg.num_layers, embed_dim=cfg.embed_dim, ffn_embed_dim=cfg.ffn_embed_dim,
                           attention_heads=cfg.attention_heads, species_vocab_size=cfg.species_vocab_size)
    m = ta.ProteinBertModel(args, ta.Alphabet.from_architecture("CodonModel"))
    m.load_state_dict(seeded_state_dict(cfg, 61, perturb=True), strict=True)
    return m.to("cuda").train().set_compute_dtype(torch.float32)


def make_batch(rank, micro):
    rng = np.random.default_rng(1000 + 10 * rank + micro)
    B, T = 3, 40
    tok = rng.integers(5, 9, (B, T))
    tok[:, 0] = 0
    labels = np.where(rng.random((B, T)) < 0.25, tok, -100)
    labels[:, 0] = -100
    sp = rng.integers(0, O.TINY.species_vocab_size, B)
    return torch.from_numpy(tok).cuda(), torch.from_numpy(sp).cuda(), torch.from_numpy(labels).cuda()


def test_two_rank_gradient_mean_matches_single_process(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = RANK_SCRIPT.format(root=ROOT, out=str(tmp_path))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", script], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=150)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    got = [torch.load(tmp_path / f"grad{r}.pt", weights_only=True) for r in range(2)]
    assert torch.equal(got[0], got[1])                      # both ranks end with the same mean
    # single process: sum over each rank's two micro-batches, then the mean over the two ranks
    want = None
    for r in range(2):
        m = make_model()
        flat = TR.FlatParameters(m)
        for micro in range(2):
            tok, sp, labels = make_batch(r, micro)
            out = m(tok, sp)
            TR.mlm_loss(out["logits"], labels).backward()
        g = flat.grad.detach().cpu().clone()
        want = g if want is None else want + g
    want /= 2
    assert float(want.abs().max()) > 0
    np.testing.assert_allclose(got[0].numpy(), want.numpy(), rtol=2e-4, atol=1e-7)
