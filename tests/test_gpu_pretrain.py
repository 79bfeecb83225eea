"""Device-side masked-LM pipeline and the pretraining loop (pretraining.py objective, gradient
accumulation, periodic checkpoints, resume) on the GPU."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import transcodon_ref as O
from transcodon_amd import pretrain
from transcodon_amd.dataset import SequenceDataset
from transcodon_amd.pipeline import DevicePipeline

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_device_pipeline_follows_the_reference_blocks():
    """Counts and invariants of DataCollator -> DataTrimmer -> DataPadder -> DataPreprocessor
    (pipeline.py:164-205, 235-263, 292-319, 330-364) on a batch with rows shorter and longer than
    max_positions; draws are from another generator than numpy's global one (statistical parity)."""
    rng = np.random.default_rng(4)
    ds = SequenceDataset()
    codons = [20, 33, 70, 130, 12]
    for m in codons:
        seq = "".join(rng.choice(list("ACGT"), 3 * m))
        ds.add(seq, int(rng.integers(0, 1438)), dot_bracket="".join(rng.choice(list(".()"), 3 * m)))
    max_pos = 120
    pipe = DevicePipeline(DEV, max_positions=max_pos, seed=3)
    items = [ds[i] for i in range(len(ds))]
    out = pipe.collate(items, step=17)
    inp, lab, st = out["input"].cpu().numpy(), out["labels"].cpu().numpy(), out["struct_label"].cpu().numpy()
    start = out["crop_start"].cpu().numpy()
    lens = np.array([3 * m + 2 for m in codons])
    assert inp.shape == (5, max_pos) and inp.dtype == np.int32 and out["struct_present"]
    assert out["organism"].dtype == torch.int64 and out["organism"].tolist() == [it["organism"] for it in items]
    for b, n in enumerate(lens):
        keep = min(n, max_pos)
        assert out["lens"][b] == keep
        assert (inp[b, keep:] == O.PAD).all() and (lab[b, keep:] == -100).all() and (st[b, keep:] == -100).all()
        s0 = int(start[b])
        assert (s0 == 0) if n <= max_pos else (0 <= s0 <= n - max_pos)
        orig = items[b]["sequence"][s0:s0 + keep]
        orig_st = items[b]["struct_label"][s0:s0 + keep]
        sel = lab[b, :keep] != -100
        assert (lab[b, :keep][sel] == orig[sel]).all()                      # labels = ground truth at drawn positions
        assert (inp[b, :keep][~sel] == orig[~sel]).all()                    # untouched elsewhere
        assert (st[b, :keep][~sel] == -100).all() and (st[b, :keep][sel] == orig_st[sel]).all()
        changed = int(n * 0.25)
        if n <= max_pos:
            assert sel.sum() == changed and not sel[0] and not sel[keep - 1]
            assert (inp[b, :keep][sel] == 9).sum() == int(changed * 0.8)
        else:                                                                # a window of a masked row
            assert 0 < sel.sum() <= changed
        assert np.isin(inp[b, :keep][sel], (5, 6, 7, 8, 9)).all()
    again = pipe.collate(items, step=17)
    assert torch.equal(again["input"], out["input"]) and torch.equal(again["labels"], out["labels"])
    other = pipe.collate(items, step=18)
    assert not torch.equal(other["labels"], out["labels"])
    # no structure labels at all: the term is reported absent (pretraining.py:115)
    plain = SequenceDataset.synthetic(3, seed=1, max_codons=40)
    assert not pipe.collate([plain[i] for i in range(3)], 0)["struct_present"]


def _args(tmp_path, extra=()):
    return pretrain.build_parser().parse_args(
        ["--synthetic", "96", "--synthetic_max_codons", "40", "--batch_size", "4", "--accumulate_gradients", "2",
         "--num_layers", "2", "--embed_dim", "128", "--ffn_embed_dim", "512", "--attention_heads", "2",
         "--species_vocab_size", "16", "--precision", "fp32", "--learning_rate", "2e-3", "--num_steps", "24",
         "--warmup_steps", "4", "--output_dir", str(tmp_path), "--seed", "5", *extra])


def test_pretraining_loop_reduces_loss_and_resumes(tmp_path):
    """20 optimiser steps of 2 accumulated micro-batches on the tiny model reduce the masked-LM loss;
    a run resumed from the step-12 checkpoint reproduces the loss of step 13."""
    args = _args(tmp_path)
    ds = SequenceDataset.synthetic(args.synthetic, seed=args.seed, max_codons=args.synthetic_max_codons, n_species=16)
    tr = pretrain.Trainer(args, ds, torch.device(DEV))
    recs = []
    for _ in range(12):
        recs.append(tr.train_step())
    path = tr.save(str(tmp_path))
    assert os.path.basename(path) == "latest-12.ckpt"
    for _ in range(8):
        recs.append(tr.train_step())
    losses = [r["train_mask_loss"] for r in recs]
    assert recs[0]["step"] == 1 and recs[-1]["step"] == 20 and "train_total_loss" not in recs[0]
    assert losses[0] > 2.0 and np.mean(losses[-3:]) < losses[0] - 0.4, losses
    assert recs[0]["lr"] == pytest.approx(2e-3 * 1 / 4) and recs[3]["lr"] == pytest.approx(2e-3)
    assert tr.validate() is not None
    # the checkpoint is in the reference's key layout (Lightning-style state_dict under "model.")
    from transcodon_amd import checkpoint
    raw = checkpoint.read_state_dict(path)
    assert all(k.startswith("model.") for k in raw) and "model.lm_head.weight" in raw
    tr2 = pretrain.Trainer(_args(tmp_path), ds, torch.device(DEV))
    tr2.load(path)
    assert (tr2.global_step, tr2.micro_step) == (12, 24)
    r13 = tr2.train_step()
    assert r13["step"] == 13 and r13["lr"] == pytest.approx(recs[12]["lr"])
    assert r13["train_mask_loss"] == pytest.approx(recs[12]["train_mask_loss"], abs=2e-4)


def test_pretrain_cli_logs_reference_metric_names(tmp_path, capsys):
    pretrain.main(["--synthetic", "40", "--synthetic_max_codons", "30", "--batch_size", "2", "--accumulate_gradients",
                   "3", "--num_layers", "1", "--embed_dim", "128", "--ffn_embed_dim", "256", "--attention_heads", "2",
                   "--species_vocab_size", "8", "--num_steps", "4", "--save_every", "2", "--val_check_interval", "2",
                   "--limit_val_batches", "1.0", "--output_dir", str(tmp_path / "run")])
    lines = [json.loads(l) for l in capsys.readouterr().out.splitlines() if l.startswith("{")]
    steps = [l for l in lines if "step" in l]
    assert [l["step"] for l in steps] == [1, 2, 3, 4]
    assert all("train_mask_loss" in l and "lr" in l for l in steps)
    assert sorted(os.listdir(tmp_path / "run")) == ["latest-2.ckpt", "latest-2.ckpt.optim", "latest-4.ckpt",
                                                    "latest-4.ckpt.optim", "metrics.jsonl"]
    assert lines[0]["event"] == "start" and lines[-1]["event"] == "done"


def test_fine_tuning_starts_from_weights_only(tmp_path):
    """--pretrained_model (fintune.py:246-248): the file's tensors replace the initial weights, keys it lacks keep
    theirs, the optimiser and the step count start fresh; the first step's loss is the trained model's, not a new one's."""
    from safetensors.torch import save_file
    from transcodon_amd import checkpoint
    args = _args(tmp_path)
    ds = SequenceDataset.synthetic(args.synthetic, seed=args.seed, max_codons=args.synthetic_max_codons, n_species=16)
    tr = pretrain.Trainer(args, ds, torch.device(DEV))
    first = tr.train_step()["train_mask_loss"]
    for _ in range(15):
        tr.train_step()
    weights = str(tmp_path / "trained.safetensors")
    checkpoint.save(tr.model, weights)
    # a second file without the structure head: those keys must be reported missing and stay at their initial values
    partial = {k: v for k, v in checkpoint.read_state_dict(weights).items() if "struct" not in k.lower()}
    assert len(partial) < len(checkpoint.read_state_dict(weights))
    save_file({k: v.clone() for k, v in partial.items()}, str(tmp_path / "partial.safetensors"))

    tr2 = pretrain.Trainer(_args(tmp_path, ["--seed", "5"]), ds, torch.device(DEV))
    before = {k: v.detach().clone() for k, v in tr2.model.state_dict().items()}
    missing, unexpected = tr2.init_from(str(tmp_path / "partial.safetensors"))
    assert unexpected == [] and missing and all("struct" in k.lower() for k in missing)
    after = tr2.model.state_dict()
    trained = tr.model.state_dict()
    for k in after:
        if k in missing:
            assert torch.equal(after[k], before[k]), k
        else:
            assert torch.equal(after[k], trained[k]), k
    assert (tr2.global_step, tr2.micro_step, tr2.opt.step_count) == (0, 0, 0)
    assert float(tr2.opt.exp_avg.abs().max()) == 0.0
    rec = tr2.train_step()
    assert rec["step"] == 1 and rec["lr"] == pytest.approx(2e-3 * 1 / 4)
    assert rec["train_mask_loss"] < first - 0.3, (rec, first)
    # the command line refuses the two starting points together
    with pytest.raises(SystemExit):
        pretrain.main(["--synthetic", "8", "--resume", "a.ckpt", "--pretrained_model", "b.safetensors",
                       "--output_dir", str(tmp_path / "x")])
