"""Masked-LM pretraining loop on the MI355X kernels.

    python -m transcodon_amd.pretrain --train_data data.csv --output_dir checkpoints/run [...]
    python -m torch.distributed.run --nproc-per-node 8 -m transcodon_amd.pretrain ...      (data parallel)

The reference drives ``CodonModel`` with a PyTorch-Lightning Trainer (pretraining.py:162-254).  This is
the same step without Lightning: flags and defaults of pretraining.py:166-192, the objective of
``training_step`` (:94-134: masked-LM cross-entropy + 0.4 x structure cross-entropy when labels exist),
AdamW + warmup-cosine of ``configure_optimizers`` (:64-92), ``accumulate_grad_batches`` micro-steps
per optimiser step with the gradient exchange only on the last one (Lightning's ``no_sync``, :242),
one process per GPU with an RCCL gradient mean (DDPStrategy, :243), a checkpoint every
``--save_every`` optimiser steps named ``latest-<step>.ckpt`` (checkpointing.py:16-26) and the scalars
Lightning logged (``train_mask_loss``, ``train_2d_mask_loss``, ``train_total_loss``, ``val_mask_loss``,
learning rate) as JSON lines on stdout and in ``<output_dir>/metrics.jsonl``.
bf16 operands with fp32 accumulation replace the reference's fp16 autocast + GradScaler.
"""

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

from . import checkpoint
from . import train as TR
from .alphabet import Alphabet
from .config import init_weights
from .dataset import SequenceDataset, train_val_split
from .model import ProteinBertModel
from .parallel import GradientSync
from .pipeline import DevicePipeline


def build_parser():
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("--train_data", type=str, default="test.csv")
    p.add_argument("--output_dir", type=str, default="checkpoints/pretain_model")
    p.add_argument("--max_positions", type=int, default=2048)
    p.add_argument("--batch_size", type=int, default=3)
    p.add_argument("--accumulate_gradients", type=int, default=6)
    p.add_argument("--mask_proportion", type=float, default=.25)
    p.add_argument("--leave_percent", type=float, default=.1)
    p.add_argument("--mask_percent", type=float, default=.8)
    p.add_argument("--warmup_steps", type=int, default=None, help="default: 10 %% of num_steps (pretraining.py:202)")
    p.add_argument("--weight_decay", type=float, default=0.1)
    p.add_argument("--lr_scheduler", type=str, default="warmup_cosine", choices=("warmup_cosine", "none"))
    p.add_argument("--learning_rate", type=float, default=2e-4)
    p.add_argument("--max_epochs", type=int, default=5)
    p.add_argument("--num_steps", type=int, default=None,
                   help="optimiser steps; default n_sequences * max_epochs / (batch_size * ranks * accumulate_gradients), "
                        "the formula of pretraining.py:201 with the actual table size")
    p.add_argument("--num_gpus", type=int, default=None, help="ignored: the rank count comes from the launcher")
    # additions
    p.add_argument("--save_every", type=int, default=20000, help="optimiser steps between checkpoints")
    p.add_argument("--val_check_interval", type=int, default=1000, help="optimiser steps between validation passes")
    p.add_argument("--limit_val_batches", type=float, default=0.1)
    p.add_argument("--resume", type=str, default=None, help="latest-<step>.ckpt to continue from")
    p.add_argument("--pretrained_model", type=str, default=None,
                   help="start from these weights (.safetensors / .ckpt) with a fresh optimiser and step count; keys the "
                        "file lacks keep their initial values (fintune.py:246-248)")
    p.add_argument("--precision", choices=("bf16", "fp32"), default="bf16")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--synthetic", type=int, default=0, help="use N random coding sequences instead of --train_data")
    p.add_argument("--synthetic_max_codons", type=int, default=400)
    p.add_argument("--log_every", type=int, default=1)
    ProteinBertModel.add_args(p)
    return p


def _dist_setup():
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    return world, rank, dev


class Trainer:
    """State of one training run: model on the flat parameter store, optimiser, data order, step count."""

    def __init__(self, args, dataset, device, world=1, rank=0):
        self.args, self.device, self.world, self.rank = args, device, world, rank
        alphabet = Alphabet.from_architecture("CodonModel")
        model = ProteinBertModel(args, alphabet)
        init_weights(model, seed=args.seed)                              # pretraining.py:46-52
        self.model = model.to(device).train().set_compute_dtype(
            torch.bfloat16 if args.precision == "bf16" else torch.float32)
        self.model.check_indices = False
        self.flat = TR.FlatParameters(self.model)
        self.opt = TR.AdamW(self.flat, lr=args.learning_rate, weight_decay=args.weight_decay)
        self.flat.grad_sync = GradientSync(self.flat)
        self.struct_params = TR.structure_head_parameters(self.model)
        self.dataset = dataset
        self.train_idx, self.val_idx = train_val_split(len(dataset), 0.003, 42)      # data_module.py:42-45
        self.pipeline = DevicePipeline(device, args.max_positions, args.mask_proportion, args.mask_percent,
                                       args.leave_percent, seed=args.seed)
        per_step = args.batch_size * world * args.accumulate_gradients
        if args.num_steps is None:
            args.num_steps = max(1, int(len(self.train_idx) * args.max_epochs / per_step))
        if args.warmup_steps is None:
            args.warmup_steps = max(1, int(args.num_steps * 0.1))
        self.global_step = 0                                               # optimiser steps done
        self.micro_step = 0                                                # micro-batches consumed by this rank

    # -- data order: one seeded permutation per epoch, ranks take interleaved batches --------------------
    def _batch_indices(self, micro_step):
        bs, n = self.args.batch_size, len(self.train_idx)
        per_epoch = max(1, n // (bs * self.world))
        epoch, k = divmod(micro_step, per_epoch)
        perm = np.random.default_rng((self.args.seed, epoch, 7)).permutation(n)
        first = (k * self.world + self.rank) * bs
        return self.train_idx[perm[first:first + bs]]

    def lr_factor(self, step):
        if self.args.lr_scheduler == "none":
            return 1.0
        return TR.warmup_cosine(step, self.args.warmup_steps, self.args.num_steps)

    def train_step(self):
        """One optimiser step = accumulate_gradients micro-batches (pretraining.py:94-134, :242)."""
        A = self.args.accumulate_gradients
        self.flat.zero_grad()
        sums = torch.zeros(3, device=self.device)                          # mask loss, 2d loss, total
        any_struct = False
        n_struct = 0                                                       # micro-batches that carried structure labels
        for a in range(A):
            items = [self.dataset[int(i)] for i in self._batch_indices(self.micro_step)]
            batch = self.pipeline.collate(items, self.micro_step * self.world + self.rank)
            self.micro_step += 1
            any_struct |= batch["struct_present"]
            last = a == A - 1
            ctx = self.flat.grad_sync.no_sync() if not last else _Null()
            with ctx:
                out = self.model(batch["input"], batch["organism"])
                mask_loss = TR.mlm_loss(out["logits"], batch["labels"])
                if batch["struct_present"]:
                    total = TR.mlm_loss(out["logits"], batch["labels"], out["struct_label_output"],
                                        batch["struct_label"], struct_present=True)
                else:
                    total = mask_loss
                (total / A).backward()
            with torch.no_grad():
                sums[0] += mask_loss.detach()
                sums[2] += total.detach()
                if batch["struct_present"]:
                    n_struct += 1
                    sums[1] += (total.detach() - mask_loss.detach()) / 0.4
        if self.world > 1:
            # The gradient mean covers the structure head on every rank, so every rank must take the same
            # decision about updating it: labels seen by any rank count for all (DDP hands the averaged
            # gradient of a parameter to the ranks that did not use it, pretraining.py:243).
            import torch.distributed as dist
            if dist.is_initialized():
                flag = torch.tensor([1.0 if any_struct else 0.0], device=self.device)
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                any_struct = bool(flag.item() > 0)
        factor = self.lr_factor(self.global_step)
        self.opt.step(lr_factor=factor, exclude=() if any_struct else self.struct_params)
        self.global_step += 1
        vals = sums.tolist()                                               # the one host synchronisation of the step
        rec = {"step": self.global_step, "train_mask_loss": vals[0] / A, "lr": self.args.learning_rate * factor}
        if n_struct:
            # the reference logs the 2d term per batch that has labels (pretraining.py:128): mean over those
            rec["train_2d_mask_loss"] = vals[1] / n_struct
            rec["train_total_loss"] = vals[2] / A
        return rec

    def validate(self):
        """val_mask_loss over limit_val_batches of the held-out rows (pretraining.py:136-160)."""
        bs = self.args.batch_size
        n_batches = int(len(self.val_idx) // bs * self.args.limit_val_batches) or (1 if len(self.val_idx) >= 1 else 0)
        if n_batches == 0:
            return None
        self.model.eval()
        tot = 0.0
        with torch.no_grad():
            for k in range(n_batches):
                idx = self.val_idx[k * bs:(k + 1) * bs]
                batch = self.pipeline.collate([self.dataset[int(i)] for i in idx], 10 ** 9 + k)
                out = self.model(batch["input"], batch["organism"])
                tot += float(TR.mlm_loss(out["logits"], batch["labels"]))
        self.model.train()
        return tot / n_batches

    # -- checkpoints (checkpointing.py:16-26 names; weights in the reference's key layout) ---------------
    def save(self, out_dir):
        path = os.path.join(out_dir, f"latest-{self.global_step}.ckpt")
        checkpoint.save(self.model, path, step=self.global_step)
        torch.save({"global_step": self.global_step, "micro_step": self.micro_step, "adam_step": self.opt.step_count,
                    "exp_avg": self.opt.exp_avg.cpu(), "exp_avg_sq": self.opt.exp_avg_sq.cpu(),
                    "num_steps": self.args.num_steps, "warmup_steps": self.args.warmup_steps},
                   path + ".optim")
        return path

    def init_from(self, path):
        """Weights only, non-strict, optimiser state and step count untouched (fintune.py:246-248).  Returns the
        (missing, unexpected) key lists of the load."""
        res = checkpoint.load_into(self.model, path, strict=False)
        return list(res.missing_keys), list(res.unexpected_keys)

    def load(self, path):
        checkpoint.load_into(self.model, path)
        st = torch.load(path + ".optim", map_location="cpu", weights_only=True)
        self.global_step, self.micro_step = int(st["global_step"]), int(st["micro_step"])
        self.opt.step_count = int(st["adam_step"])
        self.opt.exp_avg.copy_(st["exp_avg"])
        self.opt.exp_avg_sq.copy_(st["exp_avg_sq"])


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.resume and args.pretrained_model:
        raise SystemExit("--resume continues a run with its own weights; --pretrained_model starts a new one: give one of them")
    world, rank, dev = _dist_setup()
    if args.synthetic:
        ds = SequenceDataset.synthetic(args.synthetic, seed=args.seed, max_codons=args.synthetic_max_codons,
                                       n_species=args.species_vocab_size)
    else:
        ds = SequenceDataset(args.train_data)
    tr = Trainer(args, ds, dev, world, rank)
    if args.resume:
        tr.load(args.resume)
    elif args.pretrained_model:
        missing, unexpected = tr.init_from(args.pretrained_model)
        if rank == 0:
            print(json.dumps({"event": "init_from", "path": args.pretrained_model, "missing_keys": missing,
                              "unexpected_keys": unexpected}), flush=True)
    log = None
    if rank == 0:
        os.makedirs(args.output_dir, exist_ok=True)
        log = open(os.path.join(args.output_dir, "metrics.jsonl"), "a")
        print(json.dumps({"event": "start", "sequences": len(ds), "ranks": world, "num_steps": args.num_steps,
                          "warmup_steps": args.warmup_steps, "resume_step": tr.global_step}), flush=True)
    t0 = time.time()
    while tr.global_step < args.num_steps:
        rec = tr.train_step()
        if args.val_check_interval and tr.global_step % args.val_check_interval == 0:
            v = tr.validate()
            if v is not None:
                rec["val_mask_loss"] = v
        if rank == 0:
            if tr.global_step % args.log_every == 0 or "val_mask_loss" in rec:
                line = json.dumps(rec)
                print(line, flush=True)
                log.write(line + "\n")
                log.flush()
            if args.save_every and tr.global_step % args.save_every == 0:
                print(json.dumps({"event": "checkpoint", "path": tr.save(args.output_dir)}), flush=True)
    if rank == 0:
        print(json.dumps({"event": "done", "steps": tr.global_step, "hours": (time.time() - t0) / 3600}), flush=True)
        log.close()
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main())
