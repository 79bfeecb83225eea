#!/usr/bin/env python3
"""Throughput of the TransCodon forward + greedy codon decode on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3|4]

One step = one pass of the hot path over one synthetic batch per GPU.
  --config 2 (default, the headline): 256 sequences x 512 tokens per GPU (<cls> + 5'UTR + <sep> +
      masked CDS template + <eos>, no padding; species ids uniform), weak scaling.
  --config 3: 4096 sequences x 1024 tokens in all, rank r takes sequences
      [r*4096/N, (r+1)*4096/N) in micro-batches of 256 (strong scaling); a step is one pass over
      the rank's whole shard.
  --config 4 (= --mode train): masked-LM optimiser step (forward, loss, backward, gradient mean
      over ranks, AdamW).
bf16 operands with fp32 accumulation, weights random-initialised like the reference's training
module (no checkpoint is available offline).  The step covers the embedding gather, 12 encoder
layers, the final LayerNorm, the LM and structure heads and the synonymous-codon decode kernel;
inputs are resident in HBM before the clock starts.

Ranks: under torchrun (WORLD_SIZE set) this process is one rank.  Without it, --gpus N > 1 makes
this process a launcher: before anything touches the GPU it starts N copies of itself, one per
GPU, with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, and returns their exit
status.  Every rank runs its own batch (no data-path collective), the slowest rank's time is used.

Rank 0 prints ONE JSON line.  `value` comes from steps timed with the per-kernel timers OFF, the
forward of each full micro-batch replayed from one captured HIP graph (the same kernels with the
same arguments as the eager call, recorded once by ProteinBertModel.capture_forward; --no-graph
launches them one by one, and `config.eager_ms_per_step` always reports that form); a further pass
of eager steps with the timers on gives the per-kernel numbers.  `roofline`
describes the dominant kernel family (the MFMA GEMM): achieved = algorithmic FLOPs (2*M*N*K
summed over its launches) / summed device time of those launches, from HIP events attached to
every GEMM dispatch; `roofline_by_kernel` has the same for the other families.  `cpu_baseline` is
the CPU oracle (torch fp32, the reference's op sequence) timed on a bounded sample of the same
workload on the host cores.
"""

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import transcodon_amd as ta  # noqa: E402
from transcodon_amd import _lib, ops  # noqa: E402

B_PER_GPU, SEQ_LEN = 256, 512
PEAK_BF16_TFLOPS = 2500.0      # dense MFMA bf16, MI355X
PEAK_HBM_GBS = 8000.0          # HBM3E
TRAFFIC_UNIT = ("L2 fill + write-back bytes per GEMM launch (TCC_EA read / write requests of a rocprofv3 --pmc "
                "pass of this command, committed under profiles/; Infinity-Cache hits are counted in it)")
FLOPS_PER_TOKEN = lambda T: 12 * (14155776 + 3072 * T) + 2379264   # forward, E=768 L=12


def synthetic_batch(B, T, seed):
    """tokens int32 [B,T], species int32 [B], decode metadata (aa codes, offsets, first CDS row)."""
    rng = np.random.default_rng(seed)
    aas = "ACDEFGHIKLMNPQRSTVWY"
    tok = np.empty((B, T), np.int32)
    codes, offs, row0 = [], [0], []
    base_id = {"A": 5, "U": 6, "C": 7, "G": 8, "*": 9}
    for b in range(B):
        n_cds = T - 3                                   # <cls> utr <sep> cds <eos>
        m = None
        for u in [int(rng.integers(20, 101))] + list(range(20, 101)):
            if (n_cds - u) % 3 == 0:
                m = (n_cds - u) // 3
                break
        utr = rng.integers(5, 9, u)
        prot = "".join(rng.choice(list(aas), m - 1)) + "_"
        tmpl = ta.AA_tokenize(prot)
        row = [0] + list(utr) + [4] + [base_id[c] for c in tmpl] + [2]
        assert len(row) == T
        tok[b] = row
        codes.append(ops.encode_amino_acids(prot))
        offs.append(offs[-1] + m)
        row0.append(b * T + 1 + u + 1)
    species = rng.integers(0, 1438, B).astype(np.int32)
    return tok, species, b"".join(codes), np.array(offs, np.int32), np.array(row0, np.int32)


def cpu_baseline(sd_cpu, tok, species, codes_per_seq, threads):
    """Oracle forward + decode on a bounded sample (first sequences of the batch)."""
    from oracle import transcodon_ref as O
    torch.set_num_threads(threads)
    n_seq, iters = 8, 16
    t = torch.from_numpy(tok[:n_seq].astype(np.int64))
    s = torch.from_numpy(species[:n_seq].astype(np.int64))
    with torch.no_grad():
        O.forward(sd_cpu, O.FULL, t[:1], s[:1])        # warm-up
        t0 = time.perf_counter()
        for _ in range(iters):
            out = O.forward(sd_cpu, O.FULL, t, s)
            for b in range(n_seq):
                aa, start = codes_per_seq[b]
                O.decode(out["logits"][b, start:start + 3 * len(aa)], aa)
        dt = time.perf_counter() - t0
    return {"value": round(n_seq * t.shape[1] * iters / dt, 1), "unit": "tokens/s", "cores": threads,
            "host_cpu_count": os.cpu_count(), "kind": "port",
            "sample": f"{n_seq} seqs x {t.shape[1]} tokens x {iters} iters, torch fp32 oracle forward + decode, {dt:.1f} s"}


def bench_train(args, world, rank, dev, dist_on, rccl_ranks=1):
    """One step = forward, masked-LM loss, backward, gradient all-reduce (N > 1) and AdamW over
    one micro-batch per GPU (pretraining.py:94-134 with accumulate_grad_batches = 1)."""
    import torch.distributed as dist
    from transcodon_amd import train as TR
    from transcodon_amd.parallel import GradientSync
    B, T = args.train_batch, args.train_seq_len
    model = ta.ProteinBertModel(ta.default_args(), ta.Alphabet.from_architecture("CodonModel"))
    ta.init_weights(model, seed=1236)
    model = model.to(dev).train().set_compute_dtype(torch.bfloat16)
    model.check_indices = False
    flat = TR.FlatParameters(model)
    opt = TR.AdamW(flat, lr=2e-4, weight_decay=0.1)
    flat.grad_sync = GradientSync(flat)       # per-layer all-reduce overlapped with the backward (no-op at N=1)
    struct_params = TR.structure_head_parameters(model)
    tok_np, sp_np, _, _, _ = synthetic_batch(B, T, seed=1236 + rank)
    rng = np.random.default_rng(1313 + rank)
    tok_np = np.where(tok_np == 9, rng.integers(5, 9, tok_np.shape), tok_np).astype(np.int32)   # natural codons
    tok_clean, sp = torch.from_numpy(tok_np).to(dev), torch.from_numpy(sp_np).to(dev)
    lens = torch.full((B,), T, dtype=torch.int32, device=dev)
    losses = []
    struct = None
    if args.struct_labels:
        # the variant with '.()' structure labels on half of the positions: the 0.4-weighted second
        # cross-entropy term (pretraining.py:117-126) and the structure head's backward run as well
        st_np = np.where(rng.random((B, T)) < 0.5, rng.integers(0, 3, (B, T)), -100)
        struct = torch.from_numpy(st_np).to(dev)
    with_opt = [True]

    def step(i):
        # fresh corruption of the clean batch every step, on the device (the collator's job)
        tok, labels, _ = TR.mlm_mask(tok_clean, lens, seed=(1236 + rank) * 1000003 + i)
        flat.zero_grad()
        out = model(tok, sp)
        if struct is None:
            loss = TR.mlm_loss(out["logits"], labels)    # structure labels are all -100 as shipped
        else:
            loss = TR.mlm_loss(out["logits"], labels, out["struct_label_output"], struct)
        loss.backward()                                  # includes the gradient mean over ranks
        if with_opt[0]:
            opt.step(lr_factor=TR.warmup_cosine(i, 100, 10000), exclude=() if struct is not None else struct_params)
        return loss

    lib = _lib.load()
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()

    def timed(first_step, sync_grads=True):
        out = []
        flat.grad_sync.enabled = sync_grads
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            out.append(step(first_step + i))
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if dist_on:
            dist.barrier()
            tmax = torch.tensor([el], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el = float(tmax.item())
        flat.grad_sync.enabled = True
        return el, out

    lib.tc_prof_enable(0)
    elapsed, losses = timed(args.warmup)                        # the headline: no per-kernel timers
    lib.tc_prof_reset()
    lib.tc_prof_enable(1)
    elapsed_prof, _ = timed(args.warmup + args.steps)           # instrumented pass
    lib.tc_prof_enable(0)
    n, ms, flops = _lib.prof_collect(_lib.PROF_GEMM)
    an, ams, aflops = _lib.prof_collect(_lib.PROF_ATTN)
    with_opt[0] = False                                         # forward + loss + backward only
    elapsed_noopt, _ = timed(args.warmup + 2 * args.steps)
    with_opt[0] = True
    exposed_ms = None
    if dist_on:
        # the same steps without the gradient exchange (ranks drift apart: timing only, run last)
        elapsed_local, _ = timed(args.warmup + 3 * args.steps, sync_grads=False)
        exposed_ms = round(1e3 * (elapsed - elapsed_local) / args.steps, 3)
    if rank == 0:
        tokens_total = world * B * T * args.steps
        gemm_tflops = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        # forward + data gradients + weight gradients of every linear layer (3x), attention
        # forward 2 and backward 7 matrix products of 2*T*64 flops per token, head and layer
        step_flops = 3 * (12 * 14155776 + 2379264) + 12 * 3072 * T * 4.5
        model_tflops = tokens_total / world * step_flops / elapsed / 1e12
        pmc = latest_profile("pmc_traffic_train")
        result = {
            "metric": "MLM training tokens/sec (fwd+bwd+AdamW)", "value": round(tokens_total / elapsed, 1),
            "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"configs[3]: masked-LM optimiser step, {B} seqs x {T} tokens per GPU, 25 % "
                                   "positions selected (80/10/10) on the device each step, AdamW lr 2e-4 wd 0.1, gradient mean over ranks",
                       "seqs_per_gpu": B, "seq_len": T, "parallelism": f"data-parallel x{world}",
                       "rccl_ranks": rccl_ranks, "backend": args.backend if world > 1 else None,
                       "gradient_mean_bytes_per_step": int(flat.grad.numel()) * 4 if world > 1 else 0,
                       "exposed_communication_ms_per_step": exposed_ms,
                       "with_kernel_timers_ms_per_step": round(1e3 * elapsed_prof / args.steps, 3),
                       "without_adamw_ms_per_step": round(1e3 * elapsed_noopt / args.steps, 3),
                       "structure_labels": bool(args.struct_labels),
                       "first_loss": round(float(losses[0].detach()), 4), "last_loss": round(float(losses[-1].detach()), 4)},
            "roofline": {"bound": "mfma", "kernel": "gemm256x_kernel / gemm_tn256x_kernel (all GEMM launches of the step)",
                         "achieved": round(gemm_tflops, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(gemm_tflops / PEAK_BF16_TFLOPS, 4),
                         "traffic": pmc_traffic(pmc) if pmc else None,
                         "mfma_busy_pmc": pmc_mfma_busy(pmc) if pmc else None,
                         "traffic_unit": TRAFFIC_UNIT, "traffic_source": pmc and "profiles/" + pmc,
                         "launches": n,
                         "avg_launch_us": round(1e3 * ms / max(n, 1), 2),
                         "whole_step_tflops": round(model_tflops, 2),
                         "whole_step_frac": round(model_tflops / PEAK_BF16_TFLOPS, 4)},
            "roofline_by_kernel": {"attn": {"bound": "mfma", "achieved": round(aflops / (ams * 1e-3) / 1e12, 2) if ams > 0 else None,
                                            "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                                            "frac": round(aflops / (ams * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4) if ams > 0 else None,
                                            "ms_per_step": round(ams / args.steps, 3)},
                                   "gemm": {"ms_per_step": round(ms / args.steps, 3)}},
        }
        print(json.dumps(result))
    if dist_on:
        dist.destroy_process_group()


def bench_varlen(args, world, rank, dev, dist_on, rccl_ranks=1):
    """configs[4]: embedding extraction (infer_embedding.py: representations[12], mean-pooled over the
    CDS) over variable-length coding sequences, packed without padding rows.  One step = one pass
    over this rank's shard of the sequence list in batches of at most 131072 tokens; tokens/s counts
    real (non-pad) tokens only.  Strong scaling: the list is fixed, ranks take contiguous shards of
    the length-sorted list interleaved so that every rank gets the same length mix."""
    import torch.distributed as dist
    n_total = args.n_seqs or 8192
    rng = np.random.default_rng(1237)
    aa_len = np.clip(np.exp(rng.normal(5.7, 0.6, n_total)).astype(np.int64), 30, 2048)
    lens_all = 3 * aa_len + 2                                  # <cls> cds <eos>
    species_all = rng.integers(0, 1438, n_total)
    order = np.argsort(-lens_all, kind="stable")
    mine = order[rank::world]                                  # same length mix on every rank
    model = ta.ProteinBertModel(ta.default_args(), ta.Alphabet.from_architecture("CodonModel"))
    ta.init_weights(model, seed=1237)
    model = model.to(dev).eval().set_compute_dtype(torch.bfloat16)
    model.check_indices = False
    batches, cur, ntok = [], [], 0
    for i in mine:
        if cur and ntok + lens_all[i] > 131072:
            batches.append(cur); cur, ntok = [], 0
        cur.append(i); ntok += int(lens_all[i])
    if cur:
        batches.append(cur)
    prepared = []
    for idx in batches:
        toks = []
        for i in idx:
            g = np.random.default_rng((1237, int(i)))
            toks.append(np.concatenate([[0], g.integers(5, 9, lens_all[i] - 2), [2]]).astype(np.int32))
        cu = np.concatenate([[0], np.cumsum(lens_all[idx])])
        lay = model.packed_layout(species_all[idx], cu, dev)
        prepared.append((torch.from_numpy(np.concatenate(toks)).to(dev), lay,
                         torch.from_numpy((cu[:-1] + 1).astype(np.int32)).to(dev),
                         torch.from_numpy((lens_all[idx] - 2).astype(np.int32)).to(dev)))
    my_tokens = int(lens_all[mine].sum())
    flops_step = float(sum(12 * (14155776 * int(t) + 3072 * int(t) * int(t)) for t in lens_all[mine]))   # no heads

    # Batches alternate between two streams, each with its own workspace slot: the first kernels of a batch fill
    # the CUs that the last kernels of the batch before leave idle (--streams 1: one stream, one workspace).
    lanes = [torch.cuda.Stream(device=dev) for _ in range(args.streams)] if args.streams > 1 else []

    def step(overlap=True):
        for j, (tok, lay, start, ln) in enumerate(prepared):
            if lanes and overlap:
                k = j % len(lanes)
                with torch.cuda.stream(lanes[k]), model.workspace_slot(k), torch.no_grad():
                    rep = model.forward_packed(tok, None, None, want_logits=False, layout=lay)["representations"][12]
                    pooled = ops.mean_pool_packed(rep, start, ln)
                continue
            with torch.no_grad():
                rep = model.forward_packed(tok, None, None, want_logits=False, layout=lay)["representations"][12]
            pooled = ops.mean_pool_packed(rep, start, ln)
        return pooled

    def timed(n, overlap=True):
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step(overlap)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if dist_on:
            dist.barrier()
            tmax = torch.tensor([el], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el = float(tmax.item())
        return el

    lib = _lib.load()
    for _ in range(args.warmup):
        step()
    lib.tc_prof_enable(0)
    elapsed = timed(args.steps)
    elapsed_one = timed(args.steps, overlap=False)         # the same batches one after the other on one stream
    lib.tc_prof_reset()
    lib.tc_prof_enable(1)
    elapsed_prof = timed(args.steps, overlap=False)        # per-kernel device times: one stream, so that no kernel's
    lib.tc_prof_enable(0)                                  # interval contains another stream's work
    n, ms, flops = _lib.prof_collect(_lib.PROF_GEMM)
    an, ams, aflops = _lib.prof_collect(_lib.PROF_ATTN)
    ln_n, ln_ms, ln_bytes = _lib.prof_collect(_lib.PROF_LAYERNORM)
    if rank == 0:
        tokens_total = int(lens_all.sum()) * args.steps
        gemm_tflops = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        result = {
            "metric": "codon tokens/sec (fwd)", "value": round(tokens_total / elapsed, 1), "unit": "tokens/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"configs[4]: embedding extraction over {n_total} variable-length CDS (aa length lognormal(5.7, 0.6) "
                                   "clipped to [30, 2048], seed 1237; the full configuration is --n-seqs 100000), packed "
                                   "without padding rows in batches of <= 131072 tokens, mean-pooled final states",
                       "n_seqs": n_total, "tokens": int(lens_all.sum()), "mean_len": round(float(lens_all.mean()), 1),
                       "batches_per_step_rank0": len(prepared), "streams": max(1, len(lanes)), "parallelism": f"sequence-sharded x{world}",
                       "rccl_ranks": rccl_ranks, "backend": args.backend if world > 1 else None, "seqs_per_s": round(n_total * args.steps / elapsed, 1),
                       "with_kernel_timers_ms_per_step": round(1e3 * elapsed_prof / args.steps, 3),
                       "one_stream_ms_per_step": round(1e3 * elapsed_one / args.steps, 3),
                       "layernorm": "both LayerNorms of every layer folded into the q/k/v and fc1 products (q/k/v over row panels of "
                                    "<= 256 rows per sequence), residual stream as bf16 hi/lo planes; the final one is a kernel"},
            "roofline": {"bound": "mfma", "kernel": "gemm256x_kernel / gemm256_kernel (all GEMM launches of the step)",
                         "achieved": round(gemm_tflops, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(gemm_tflops / PEAK_BF16_TFLOPS, 4), "traffic": None,
                         "traffic_source": "not collected for this configuration (no counter pass of this command is committed)",
                         "launches": n,
                         "avg_launch_us": round(1e3 * ms / max(n, 1), 2), "ms_per_step": round(ms / args.steps, 3),
                         "whole_forward_tflops": round(flops_step * args.steps / elapsed / 1e12, 2),
                         "whole_forward_frac": round(flops_step * args.steps / elapsed / 1e12 / PEAK_BF16_TFLOPS, 4)},
            "roofline_by_kernel": {
                "attn": {"bound": "mfma", "achieved": round(aflops / (ams * 1e-3) / 1e12, 2) if ams > 0 else None,
                         "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "ms_per_step": round(ams / args.steps, 3)},
                "layernorm": {"bound": "hbm", "achieved": round(ln_bytes / (ln_ms * 1e-3) / 1e9, 2) if ln_ms > 0 else None,
                              "peak": PEAK_HBM_GBS, "unit": "GB/s", "ms_per_step": round(ln_ms / args.steps, 3)}},
        }
        print(json.dumps(result))
    if dist_on:
        dist.destroy_process_group()


def pmc_mfma_busy(path, prefix="gemm"):
    """Launch-weighted share of SIMD cycles with the matrix pipe busy, from the same committed pass."""
    try:
        with open(os.path.join(ROOT, "profiles", path)) as fh:
            ks = json.load(fh)["kernels"]
    except (OSError, ValueError, KeyError):
        return None
    sel = [v for k, v in ks.items() if k.startswith(prefix) and "mfma_busy" in v]
    n = sum(v["dispatches"] for v in sel)
    return round(sum(v["mfma_busy"] * v["dispatches"] for v in sel) / n, 4) if n else None


def pmc_traffic(path, prefix="gemm"):
    """HBM bytes per launch of a kernel family from a committed counter pass (rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE collected separately on this same command): launch-weighted mean."""
    try:
        with open(os.path.join(ROOT, "profiles", path)) as fh:
            ks = json.load(fh)["kernels"]
    except (OSError, ValueError, KeyError):
        return None
    sel = [v for k, v in ks.items() if k.startswith(prefix)]
    n = sum(v["dispatches"] for v in sel)
    if not n:
        return None
    return round(sum((v["read_bytes"] + v["write_bytes"]) * v["dispatches"] for v in sel) / n)


def latest_profile(stem):
    """Newest committed counter summary profiles/rNN_<stem>.json (None if there is none)."""
    import glob
    hits = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r??_{stem}.json")))
    return os.path.basename(hits[-1]) if hits else None


def launch_ranks(n, argv):
    """Start n copies of this script, one rank per GPU, and return the worst exit status.  Runs in a
    process that has not touched the GPU (importing torch does not)."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for pr in list(pending):
                code = pr.poll()
                if code is None:
                    continue
                pending.remove(pr)
                if code != 0:
                    rc = rc or code
                    for other in pending:          # a failed rank would leave the others at a barrier
                        other.terminate()
            time.sleep(0.05)
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    return rc


def sequence_batch(idx, T, seed):
    """Sequences idx (global indices) of the config-3 workload: every sequence has its own generator
    stream, so a rank builds only its shard."""
    toks, sps, codes, offs, row0 = [], [], [], [0], []
    for k, i in enumerate(idx):
        t, sp, c, o, r0 = synthetic_batch(1, T, seed=(seed, int(i)))
        toks.append(t[0]); sps.append(sp[0]); codes.append(c)
        offs.append(offs[-1] + int(o[1])); row0.append(k * T + int(r0[0]))
    return (np.stack(toks), np.array(sps, np.int32), b"".join(codes), np.array(offs, np.int32),
            np.array(row0, np.int32))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--config", type=int, choices=(2, 3, 4, 5), default=2,
                    help="BASELINE.json configs[] entry (1-based): 2 = 256 x 512 forward (headline), "
                         "3 = 4096 x 1024 batch-sharded, 4 = masked-LM step, 5 = varlen embedding extraction")
    ap.add_argument("--mode", choices=("infer", "train"), default=None,
                    help="train = --config 4 (kept for older command lines)")
    ap.add_argument("--train-batch", type=int, default=64)
    ap.add_argument("--train-seq-len", type=int, default=1024)
    ap.add_argument("--streams", type=int, choices=(1, 2, 3, 4), default=2,
                    help="inference: HIP streams the captured forwards alternate on (2: consecutive micro-batches "
                         "overlap at their kernels' tails; each stream has its own graph and workspace)")
    ap.add_argument("--no-graph", action="store_true",
                    help="inference configs: launch the forward kernel by kernel instead of replaying the captured HIP graph")
    ap.add_argument("--no-fold", action="store_true",
                    help="inference configs: run the LayerNorms of the layers as separate kernels")
    ap.add_argument("--struct-labels", action="store_true",
                    help="config 4 with structure labels on half of the positions (the 0.4-weighted second loss term)")
    ap.add_argument("--n-seqs", type=int, default=None,
                    help="config 3 / 5: total number of sequences (default 4096 / 100000)")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="process-group backend; gloo only to rehearse the multi-rank path on a box with one GPU")
    ap.add_argument("--share-device", action="store_true",
                    help="rehearsal: every rank uses GPU 0 (RCCL refuses two ranks on one device, so use --backend gloo)")
    ap.add_argument("--dry-run", action="store_true",
                    help="print this rank's identity and exit before any GPU call (launcher test)")
    args = ap.parse_args()
    if args.mode == "train":
        args.config = 4

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1
    if args.dry_run:
        print(json.dumps({"dry_run": True, "rank": rank, "local_rank": local_rank, "world": world,
                          "master": f"{os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}",
                          "config": args.config}), flush=True)
        return
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rccl_ranks = 1
    if dist_on:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
        ones = torch.ones(1, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(ones)                      # the rank count RCCL itself sees
        rccl_ranks = int(ones.item())
        assert rccl_ranks == dist.get_world_size() == world

    if args.config == 4:
        return bench_train(args, world, rank, dev, dist_on, rccl_ranks)
    if args.config == 5:
        return bench_varlen(args, world, rank, dev, dist_on, rccl_ranks)

    model = ta.ProteinBertModel(ta.default_args(), ta.Alphabet.from_architecture("CodonModel"))
    ta.init_weights(model, seed=1234)
    sd_cpu = {k: v.detach().clone() for k, v in model.state_dict().items()} if rank == 0 else None
    model = model.to(dev).eval().set_compute_dtype(torch.bfloat16)
    model.check_indices = False          # inputs are generated in range; avoids a host sync per step
    model.fold_layernorm = not args.no_fold

    if args.config == 2:
        B, T, n_micro = B_PER_GPU, SEQ_LEN, 1
        tok_np, sp_np, codes, offs, row0 = synthetic_batch(B, T, seed=1234 + rank)
        n_total = world * B
    else:
        # config 3: strong scaling over a fixed set of sequences, micro-batches of 256
        from transcodon_amd.parallel import shard_range
        T, n_total = 1024, args.n_seqs or 4096
        lo, hi = shard_range(n_total, rank, world)
        tok_np, sp_np, codes, offs, row0 = sequence_batch(range(lo, hi), T, seed=1235)
        B = min(256, hi - lo)
        n_micro = (hi - lo + B - 1) // B
    host = {"tok": torch.from_numpy(tok_np).pin_memory(), "sp": torch.from_numpy(sp_np).pin_memory(),
            "aa": torch.frombuffer(bytearray(codes), dtype=torch.uint8).pin_memory(),
            "offs": torch.from_numpy(offs).pin_memory(), "row0": torch.from_numpy(row0).pin_memory()}
    # one set of device input buffers per stream (the step with host I/O refills the set of the stream it runs on)
    dsets = [{k: v.to(dev) for k, v in host.items()} for _ in range(args.streams if not args.no_graph else 1)]
    d = dsets[0]
    n_seq = tok_np.shape[0]
    bases_host = torch.empty(3 * len(codes), dtype=torch.uint8).pin_memory()

    # per micro-batch views / rebased offsets, made before the clock starts
    def views(ds):
        out = []
        for i in range(n_micro):
            lo_s, hi_s = i * B, min((i + 1) * B, n_seq)
            a0, a1 = int(offs[lo_s]), int(offs[hi_s])
            out.append((ds["tok"][lo_s:hi_s], ds["sp"][lo_s:hi_s], ds["aa"][a0:a1], lo_s, hi_s, a0, a1))
        return out
    # (offsets relative to the micro-batch do not change with the step: made once)
    rel = []
    for i in range(n_micro):
        lo_s, hi_s = i * B, min((i + 1) * B, n_seq)
        a0 = int(offs[lo_s])
        rel.append(((d["offs"][lo_s:hi_s + 1] - a0).contiguous(), (d["row0"][lo_s:hi_s] - lo_s * T).contiguous()))
    mbs_of = [views(ds) for ds in dsets]

    # The forward of a full micro-batch is replayed from one HIP graph (ProteinBertModel.capture_forward: the same
    # launches, recorded once); a refused capture falls back to the eager call and says so in the result.
    # With two streams every stream has its own graph and its own workspace (the graphs of one model otherwise
    # share the model's), and consecutive micro-batches alternate between them: nothing orders one against the
    # other, so the next forward's first kernels fill the CUs that the previous one's last kernels leave idle.
    graphs, graph_err = [], None
    if not args.no_graph:
        try:
            graphs = [model.capture_forward(B, T, private_workspace=args.streams > 1) for _ in range(args.streams)]
        except Exception as e:
            graphs, graph_err = [], f"{type(e).__name__}: {e}"
    graphed = graphs[0] if graphs else None
    lanes = [torch.cuda.Stream(device=dev) for _ in range(len(graphs))] if len(graphs) > 1 else []
    turn = [0]

    def micro(i, use_graph, which=0):
        """Forward + decode of micro-batch i of this rank's shard (inputs of buffer set `which`); returns the decoded bases."""
        tok_i, sp_i, aa_i, _, _, a0, a1 = mbs_of[which][i]
        offs_i, row0_i = rel[i]
        if use_graph and graphed is not None and tok_i.shape[0] == B:
            g = graphs[turn[0] % len(graphs)]
            out = g(tok_i, sp_i, check_indices=False)
        else:
            with torch.no_grad():
                out = model(tok_i, sp_i)
        return ops.codon_decode(out["logits"].view(tok_i.shape[0] * T, -1), aa_i, offs_i, row0_i), a0, a1

    def step(host_io=False, use_graph=True, overlap=True):
        laned = bool(lanes) and use_graph and overlap
        if host_io and not laned:      # token ids, species and residues come from the host; decoded bases go back
            for k in ("tok", "sp", "aa", "offs", "row0"):
                d[k].copy_(host[k], non_blocking=True)
        for i in range(n_micro):
            if laned:
                which = turn[0] % len(lanes)
                with torch.cuda.stream(lanes[which]):
                    if host_io:                            # this micro-batch's part of the inputs, on its own stream
                        _, _, _, lo_s, hi_s, a0, a1 = mbs_of[which][i]
                        ds = dsets[which]
                        ds["tok"][lo_s:hi_s].copy_(host["tok"][lo_s:hi_s], non_blocking=True)
                        ds["sp"][lo_s:hi_s].copy_(host["sp"][lo_s:hi_s], non_blocking=True)
                        ds["aa"][a0:a1].copy_(host["aa"][a0:a1], non_blocking=True)
                        ds["offs"][lo_s:hi_s + 1].copy_(host["offs"][lo_s:hi_s + 1], non_blocking=True)
                        ds["row0"][lo_s:hi_s].copy_(host["row0"][lo_s:hi_s], non_blocking=True)
                    bases, a0, a1 = micro(i, use_graph, which)
                    if host_io:
                        bases_host[3 * a0:3 * a1].copy_(bases, non_blocking=True)
                turn[0] += 1
                continue
            bases, a0, a1 = micro(i, use_graph)
            if host_io:
                bases_host[3 * a0:3 * a1].copy_(bases, non_blocking=True)
        return bases

    def timed(n, **kw):
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            last = step(**kw)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if dist_on:
            dist.barrier()
            tmax = torch.tensor([el], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el = float(tmax.item())
        return el, last

    lib = _lib.load()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    lib.tc_prof_enable(0)
    elapsed, bases = timed(args.steps)                      # the headline: no per-kernel timers
    elapsed_io, _ = timed(args.steps, host_io=True)         # the same with H2D of the inputs and D2H of the bases
    elapsed_one, bases_one = timed(args.steps, overlap=False)         # the graphs one after the other on one stream
    if not torch.equal(bases_one, bases):
        raise SystemExit("bench: the overlapped and the one-stream step decode different bases")
    elapsed_eager, bases_eager = timed(args.steps, use_graph=False)   # the same step launched kernel by kernel
    if graphed is not None and not torch.equal(bases_eager, bases):
        raise SystemExit(f"bench: the graph replay and the eager step decode different bases "
                         f"({int((bases_eager != bases).sum())} of {bases.numel()})")
    lib.tc_prof_reset()
    lib.tc_prof_enable(1)
    elapsed_prof, _ = timed(args.steps, use_graph=False)    # instrumented pass (eager launches): per-kernel device times
    lib.tc_prof_enable(0)
    prof = {name: _lib.prof_collect(kind) for name, kind in
            (("gemm", _lib.PROF_GEMM), ("attn", _lib.PROF_ATTN), ("layernorm", _lib.PROF_LAYERNORM),
             ("embed", _lib.PROF_EMBED), ("head", _lib.PROF_HEAD), ("decode", _lib.PROF_DECODE),
             ("ln_rows", _lib.PROF_LN_ROWS))}
    folded = bool(model.fold_layernorm) and lib.tc_ln_fold_supported(_lib.TC_BF16, B, T, 768, 3072) != 0

    # sanity: the decoded bases of the last micro-batch translate back to its proteins
    flat = bytes(bases.cpu().numpy()).decode("ascii")
    from transcodon_amd.sequence import CODON_TO_AA
    lo_s = (n_micro - 1) * B
    a0, a1 = int(offs[lo_s]), int(offs[lo_s + 1])
    first = "".join(CODON_TO_AA[flat[i:i + 3]] for i in range(0, 3 * (a1 - a0), 3))
    assert first.replace("#", "_") == "".join(ops.AA_ORDER[c] for c in codes[a0:a1])

    if rank == 0:
        tokens_step = n_total * T                            # whole job, all ranks, one step
        tokens_total = tokens_step * args.steps
        n, ms, flops = prof["gemm"]
        gemm_tflops = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        model_tflops = tokens_total / world * FLOPS_PER_TOKEN(T) / elapsed / 1e12
        # operand + result bytes of the GEMMs of one forward (bf16 activations / weights, f32 residual stream)
        M, E, F = B * T, 768, 3072
        per_layer = (M * E * 2 + 3 * E * E * 2 + M * 3 * E * 2) + (M * E * 2 + E * E * 2 + 2 * M * E * 4) + \
                    (M * E * 2 + F * E * 2 + M * F * 2) + (M * F * 2 + E * F * 2 + 2 * M * E * 4)
        # (with folded LayerNorms the residual stream is a pair of bf16 planes: the same 4 bytes per element
        # in and out of the residual products as the f32 stream)
        gemm_bytes = (12 * per_layer + M * E * 2 + 2 * E * E * 2 + M * 2 * E * 2) * args.steps * n_micro
        by_kernel = {}
        for name, bound, unit, scale, peak in (("attn", "mfma", "TFLOP/s", 1e12, PEAK_BF16_TFLOPS),
                                               ("layernorm", "hbm", "GB/s", 1e9, PEAK_HBM_GBS),
                                               ("embed", "hbm", "GB/s", 1e9, PEAK_HBM_GBS),
                                               ("head", "hbm", "GB/s", 1e9, PEAK_HBM_GBS),
                                               ("decode", "hbm", "GB/s", 1e9, PEAK_HBM_GBS),
                                               ("ln_rows", "hbm", "GB/s", 1e9, PEAK_HBM_GBS)):
            cnt, kms, work = prof[name]
            if kms > 0:
                ach = work / (kms * 1e-3) / scale
                by_kernel[name] = {"bound": bound, "achieved": round(ach, 2), "peak": peak, "unit": unit,
                                   "frac": round(ach / peak, 4), "launches": cnt, "avg_us": round(1e3 * kms / cnt, 2),
                                   "ms_per_step": round(kms / args.steps, 3)}
        pmc = latest_profile("pmc_traffic_infer")
        workload = ("configs[1]: 256 seqs x 512 tokens (species + 5'UTR + CDS template) per GPU, "
                    "forward + LM/structure heads + greedy codon decode") if args.config == 2 else \
                   (f"configs[2]: {n_total} seqs x {T} tokens sharded over the ranks in micro-batches of {B}, "
                    "forward + LM/structure heads + greedy codon decode")
        result = {
            "metric": "codon tokens/sec (fwd)", "value": round(tokens_total / elapsed, 1),
            "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
            "scaling": "weak" if args.config == 2 else "strong", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": workload,
                       "seqs_per_gpu": n_seq, "seq_len": T, "micro_batch": B,
                       "codons_per_s": round(tokens_total / elapsed / 3, 1),
                       "weights": "random init N(0,0.02), 89.8M params", "parallelism": f"batch-sharded x{world}",
                       "rccl_ranks": rccl_ranks, "backend": args.backend if world > 1 else None,
                       "layernorm": ("24 of 25 folded into the q/k/v, out-proj, fc1 and fc2 products, residual stream as bf16 hi/lo planes" if folded
                                     else "separate kernels"),
                       "value_definition": ("tokens/s with token ids, species and residues already resident in HBM and the "
                                            "decoded bases left there; the same steps with the H2D copy of the inputs and the "
                                            "D2H copy of the decoded bases inside the timed region are value_with_host_io"),
                       "value_with_host_io": round(tokens_total / elapsed_io, 1),
                       "with_kernel_timers_ms_per_step": round(1e3 * elapsed_prof / args.steps, 3),
                       "launch": ((f"one HIP graph per forward (ProteinBertModel.capture_forward), decode kernel eager; "
                                   f"{len(graphs)} graph(s) with private workspaces alternating on {max(1, len(lanes))} stream(s)")
                                  if graphed is not None else "eager, kernel by kernel"),
                       "one_stream_ms_per_step": round(1e3 * elapsed_one / args.steps, 3),
                       "eager_ms_per_step": round(1e3 * elapsed_eager / args.steps, 3),
                       "hip_graph_error": graph_err},
            "roofline": {"bound": "mfma", "kernel": "gemm256x_kernel<bf16, EPI> (all GEMM launches of the step)",
                         "achieved": round(gemm_tflops, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(gemm_tflops / PEAK_BF16_TFLOPS, 4),
                         "traffic": pmc_traffic(pmc) if pmc and args.config == 2 else None,
                         "mfma_busy_pmc": pmc_mfma_busy(pmc) if pmc and args.config == 2 else None,
                         "traffic_unit": TRAFFIC_UNIT,
                         "traffic_source": ("profiles/" + pmc) if pmc and args.config == 2 else
                                           "not collected for this configuration (no counter pass of this command is committed)",
                         "algorithmic_bytes_per_launch": round(gemm_bytes / max(n, 1)),
                         "launches": n, "avg_launch_us": round(1e3 * ms / max(n, 1), 2),
                         "ms_per_step": round(ms / args.steps, 3),
                         "whole_forward_tflops": round(model_tflops, 2),
                         "whole_forward_frac": round(model_tflops / PEAK_BF16_TFLOPS, 4)},
            "roofline_by_kernel": by_kernel,
        }
        if world == 1 and not args.no_cpu_baseline:
            per_seq = []
            for b in range(8):
                aa = "".join(ops.AA_ORDER[c] for c in codes[int(offs[b]):int(offs[b + 1])])
                per_seq.append((aa, int(row0[b]) - b * T))
            threads = min(os.cpu_count() or 1, 16)
            result["cpu_baseline"] = cpu_baseline(sd_cpu, tok_np, sp_np, per_seq, threads)
        print(json.dumps(result))
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
