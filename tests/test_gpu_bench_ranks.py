"""bench.py with two ranks, as a collected test: the launcher, the process group, the sharding of every
configuration and the gradient exchange of the training step run here before any multi-GPU node sees
them.  The box has one GPU, so both ranks use it (--share-device) and the process group is gloo (RCCL
refuses two ranks on one device); everything else is the code path of `bench.py --gpus N`: one process
per rank started before anything touches the GPU (pretraining.py:235-246 runs one process per GPU under
DDP), barrier + synchronize around the timed steps, the maximum over ranks, one JSON line from rank 0.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-device", "--backend", "gloo",
           "--steps", "1", "--warmup", "1", "--no-cpu-baseline", *extra]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]           # rank 0 only
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["rccl_ranks"] == 2 and rec["config"]["backend"] == "gloo"
    assert rec["steps"] == 1 and rec["warmup"] == 1 and rec["value"] > 0 and rec["ms_per_step"] > 0
    assert "cpu_baseline" not in rec
    return rec


def test_two_ranks_headline_config():
    rec = _bench("--config", "2")
    assert rec["scaling"] == "weak" and rec["config"]["parallelism"] == "batch-sharded x2"
    assert rec["config"]["seqs_per_gpu"] == 256 and rec["config"]["seq_len"] == 512
    # whole-job value: both ranks' tokens over the slowest rank's time
    assert abs(rec["value"] - 2 * 256 * 512 / (rec["ms_per_step"] * 1e-3)) <= 1e-3 * rec["value"]
    assert rec["roofline"]["frac"] > 0


def test_two_ranks_sharded_long_sequences():
    rec = _bench("--config", "3", "--n-seqs", "64")
    assert rec["scaling"] == "strong" and rec["config"]["parallelism"] == "batch-sharded x2"
    assert rec["config"]["seqs_per_gpu"] == 32 and rec["config"]["seq_len"] == 1024
    assert abs(rec["value"] - 64 * 1024 / (rec["ms_per_step"] * 1e-3)) <= 1e-3 * rec["value"]


def test_two_ranks_training_step_exchanges_gradients():
    import transcodon_amd as ta
    from transcodon_amd.train import PackedLayout
    rec = _bench("--config", "4", "--train-batch", "4", "--train-seq-len", "256")
    cfg = rec["config"]
    assert rec["scaling"] == "weak" and cfg["parallelism"] == "data-parallel x2"
    assert cfg["exposed_communication_ms_per_step"] is not None
    model = ta.ProteinBertModel(ta.default_args(), ta.Alphabet.from_architecture("CodonModel"))
    assert cfg["gradient_mean_bytes_per_step"] == 4 * PackedLayout(model).total
    assert cfg["first_loss"] > 0 and cfg["last_loss"] > 0
    assert abs(rec["value"] - 2 * 4 * 256 / (rec["ms_per_step"] * 1e-3)) <= 1e-3 * rec["value"]


def test_two_ranks_variable_length_extraction():
    rec = _bench("--config", "5", "--n-seqs", "64")
    assert rec["scaling"] == "strong" and rec["config"]["parallelism"] == "sequence-sharded x2"
    assert rec["config"]["n_seqs"] == 64
    assert abs(rec["value"] - rec["config"]["tokens"] / (rec["ms_per_step"] * 1e-3)) <= 1e-3 * rec["value"]
