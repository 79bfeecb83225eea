#!/usr/bin/env python3
"""Throughput of the TransCodon forward + greedy codon decode on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One step = one pass of the hot path over one synthetic batch per GPU: 256 sequences x 512
tokens (<cls> + 5'UTR + <sep> + masked CDS template + <eos>, no padding; species ids
uniform), bf16 operands with fp32 accumulation, weights random-initialised like the
reference's training module (no checkpoint is available offline).  The step covers the
embedding gather, 12 encoder layers, the final LayerNorm, the LM and structure heads and
the synonymous-codon decode kernel; inputs are resident in HBM before the clock starts.
For N > 1 the driver launches one process per GPU; every rank runs its own batch (weak
scaling, no data-path collective) and the slowest rank's time is used.

Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel family (the MFMA
GEMM): achieved = algorithmic FLOPs (2*M*N*K summed over its launches) / summed device time
of those launches, taken from HIP events that bracket every GEMM launch inside the timed
region.  `cpu_baseline` is the CPU oracle (torch fp32, the reference's op sequence) timed on
a bounded sample of the same workload on the host cores.
"""

import argparse
import json
import os
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
            "kind": "port",
            "sample": f"{n_seq} seqs x {t.shape[1]} tokens x {iters} iters, torch fp32 oracle forward + decode, {dt:.1f} s"}


def bench_train(args, world, rank, dev, dist_on):
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

    def step(i):
        # fresh corruption of the clean batch every step, on the device (the collator's job)
        tok, labels, _ = TR.mlm_mask(tok_clean, lens, seed=(1236 + rank) * 1000003 + i)
        flat.zero_grad()
        out = model(tok, sp)
        loss = TR.mlm_loss(out["logits"], labels)        # structure labels are all -100 as shipped
        loss.backward()                                  # includes the gradient mean over ranks
        opt.step(lr_factor=TR.warmup_cosine(i, 100, 10000), exclude=struct_params)
        return loss

    lib = _lib.load()
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    lib.tc_prof_reset()
    lib.tc_prof_enable(1)
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        losses.append(step(args.warmup + i))
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist_on:
        dist.barrier()
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    lib.tc_prof_enable(0)
    n, ms, flops = _lib.prof_collect(_lib.PROF_GEMM)
    an, ams, aflops = _lib.prof_collect(_lib.PROF_ATTN)
    if rank == 0:
        tokens_total = world * B * T * args.steps
        gemm_tflops = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        # forward + data gradients + weight gradients of every linear layer (3x), attention
        # forward 2 and backward 7 matrix products of 2*T*64 flops per token, head and layer
        step_flops = 3 * (12 * 14155776 + 2379264) + 12 * 3072 * T * 4.5
        model_tflops = tokens_total / world * step_flops / elapsed / 1e12
        result = {
            "metric": "MLM training tokens/sec (fwd+bwd+AdamW)", "value": round(tokens_total / elapsed, 1),
            "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"configs[3]: masked-LM optimiser step, {B} seqs x {T} tokens per GPU, 25 % "
                                   "positions selected (80/10/10) on the device each step, AdamW lr 2e-4 wd 0.1, gradient mean over ranks",
                       "seqs_per_gpu": B, "seq_len": T, "parallelism": f"data-parallel x{world}",
                       "first_loss": round(float(losses[0]), 4), "last_loss": round(float(losses[-1]), 4)},
            "roofline": {"bound": "mfma", "kernel": "gemm256x_kernel / gemm_tn256x_kernel (all GEMM launches of the step)",
                         "achieved": round(gemm_tflops, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(gemm_tflops / PEAK_BF16_TFLOPS, 4),
                         "traffic": pmc_traffic("r01_pmc_traffic_train.json"),
                         "mfma_busy_pmc": pmc_mfma_busy("r01_pmc_traffic_train.json"),
                         "traffic_unit": "HBM bytes per GEMM launch (PMC pass of this command, committed under profiles/)",
                         "launches": n,
                         "avg_launch_us": round(1e3 * ms / max(n, 1), 2),
                         "whole_step_tflops": round(model_tflops, 2),
                         "whole_step_frac": round(model_tflops / PEAK_BF16_TFLOPS, 4)},
            "kernels": {"gemm_ms_per_step": round(ms / args.steps, 3), "attn_ms_per_step": round(ams / args.steps, 3),
                        "attn_tflops": round(aflops / (ams * 1e-3) / 1e12, 2) if ams > 0 else None},
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", choices=("infer", "train"), default="infer",
                    help="infer: forward + decode (headline metric); train: masked-LM optimiser step")
    ap.add_argument("--train-batch", type=int, default=64)
    ap.add_argument("--train-seq-len", type=int, default=1024)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if dist_on:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    if args.mode == "train":
        return bench_train(args, world, rank, dev, dist_on)

    B, T = B_PER_GPU, SEQ_LEN
    model = ta.ProteinBertModel(ta.default_args(), ta.Alphabet.from_architecture("CodonModel"))
    ta.init_weights(model, seed=1234)
    sd_cpu = {k: v.detach().clone() for k, v in model.state_dict().items()} if rank == 0 else None
    model = model.to(dev).eval().set_compute_dtype(torch.bfloat16)
    model.check_indices = False          # inputs are generated in range; avoids a host sync per step

    tok_np, sp_np, codes, offs, row0 = synthetic_batch(B, T, seed=1234 + rank)
    tok = torch.from_numpy(tok_np).to(dev)
    sp = torch.from_numpy(sp_np).to(dev)
    aa_dev = torch.frombuffer(bytearray(codes), dtype=torch.uint8).to(dev)
    offs_dev, row0_dev = torch.from_numpy(offs).to(dev), torch.from_numpy(row0).to(dev)

    def step():
        with torch.no_grad():
            out = model(tok, sp)
        return ops.codon_decode(out["logits"].view(B * T, -1), aa_dev, offs_dev, row0_dev)

    lib = _lib.load()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    lib.tc_prof_reset()
    lib.tc_prof_enable(1)
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        bases = step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist_on:
        dist.barrier()
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    lib.tc_prof_enable(0)
    prof = {name: _lib.prof_collect(kind) for name, kind in
            (("gemm", _lib.PROF_GEMM), ("attn", _lib.PROF_ATTN), ("layernorm", _lib.PROF_LAYERNORM),
             ("embed", _lib.PROF_EMBED), ("head", _lib.PROF_HEAD), ("decode", _lib.PROF_DECODE))}

    # sanity: the decoded bases of the last step translate back to the proteins
    flat = bytes(bases.cpu().numpy()).decode("ascii")
    from transcodon_amd.sequence import CODON_TO_AA
    first = "".join(CODON_TO_AA[flat[i:i + 3]] for i in range(0, 3 * int(offs[1]), 3))
    assert first.replace("#", "_") == "".join(ops.AA_ORDER[c] for c in codes[:int(offs[1])])

    if rank == 0:
        tokens_total = world * B * T * args.steps
        n, ms, flops = prof["gemm"]
        gemm_tflops = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        model_tflops = tokens_total / world * FLOPS_PER_TOKEN(T) / elapsed / 1e12
        # operand + result bytes of the GEMMs of one forward (bf16 activations / weights, f32 residual stream)
        M, E, F = B * T, 768, 3072
        per_layer = (M * E * 2 + 3 * E * E * 2 + M * 3 * E * 2) + (M * E * 2 + E * E * 2 + 2 * M * E * 4) + \
                    (M * E * 2 + F * E * 2 + M * F * 2) + (M * F * 2 + E * F * 2 + 2 * M * E * 4)
        gemm_bytes = (12 * per_layer + M * E * 2 + 2 * E * E * 2 + M * 2 * E * 2) * args.steps
        extra = {}
        for name, unit, scale in (("attn", "TFLOP/s", 1e12), ("layernorm", "GB/s", 1e9),
                                  ("embed", "GB/s", 1e9), ("head", "GB/s", 1e9), ("decode", "GB/s", 1e9)):
            cnt, kms, work = prof[name]
            if kms > 0:
                extra[name] = {"launches": cnt, "avg_us": round(1e3 * kms / cnt, 2),
                               "achieved": round(work / (kms * 1e-3) / scale, 2), "unit": unit}
        result = {
            "metric": "codon tokens/sec (fwd)", "value": round(tokens_total / elapsed, 1),
            "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "configs[1]: 256 seqs x 512 tokens (species + 5'UTR + CDS template) per GPU, "
                                   "forward + LM/structure heads + greedy codon decode",
                       "seqs_per_gpu": B, "seq_len": T, "codons_per_s": round(tokens_total / elapsed / 3, 1),
                       "weights": "random init N(0,0.02), 89.8M params", "parallelism": f"batch-sharded x{world}"},
            "roofline": {"bound": "mfma", "kernel": "gemm256x_kernel<bf16, EPI> (all GEMM launches of the step)",
                         "achieved": round(gemm_tflops, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(gemm_tflops / PEAK_BF16_TFLOPS, 4),
                         "traffic": pmc_traffic("r01_pmc_traffic_infer.json"),
                         "mfma_busy_pmc": pmc_mfma_busy("r01_pmc_traffic_infer.json"),
                         "traffic_unit": "HBM bytes per GEMM launch (PMC pass of this command, committed under profiles/)",
                         "algorithmic_bytes_per_launch": round(gemm_bytes / max(n, 1)),
                         "launches": n, "avg_launch_us": round(1e3 * ms / max(n, 1), 2),
                         "whole_forward_tflops": round(model_tflops, 2),
                         "whole_forward_frac": round(model_tflops / PEAK_BF16_TFLOPS, 4)},
            "kernels": extra,
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
