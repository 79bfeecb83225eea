"""world_size-2 gloo tests of the sharding helpers (CPU, two real processes)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from transcodon_amd.parallel import GradientSync, allreduce_grads, distributed_map, shard_range


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 8, 9, 4096, 100003):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        items = [f"P{i}" for i in range(11)]
        seen = []

        def fn(shard):
            seen.extend(shard)
            return [s.lower() + f"@{rank}" for s in shard]

        res = distributed_map(items, fn)
        b, e = shard_range(len(items), rank, world)
        assert seen == items[b:e]
        if rank == 0:
            assert [r.split("@")[0] for r in res] == [s.lower() for s in items]
            assert [int(r.split("@")[1]) for r in res] == [0] * 6 + [1] * 5
        else:
            assert res is None
        everywhere = distributed_map(items, fn, gather_to=None)
        assert len(everywhere) == len(items)
        # empty shard on the second rank
        one = distributed_map(["only"], lambda sh: [x.upper() for x in sh])
        if rank == 0:
            assert one == ["ONLY"]
        g = torch.full((5,), float(rank + 1))
        allreduce_grads(g)
        assert torch.allclose(g, torch.full((5,), 1.5))
        # overlapped per-layer gradient mean on a packed flat store (CPU tensors, same code path
        # as the GPU one minus the side stream)
        class _Layout:
            loff = [{"a": (8, 3), "b": (12, 4)}, {"a": (16, 5), "b": (24, 2)}]
            hoff = {"st_bias": (0, 3), "st_bias2": (4, 3)}
            total = 32

        class _Store:
            layout = _Layout()

            def __init__(self):
                self.grad = torch.arange(32, dtype=torch.float32) * (rank + 1)

            def sync_shared_bias_grad(self):
                self.grad[4:7] = self.grad[0:3]

        store = _Store()
        sync = GradientSync(store)
        assert sync.layer_ranges == [(8, 16), (16, 32)] and sync.head_range == (0, 8)
        with sync.no_sync():                      # a micro-batch inside an accumulation window
            for layer in (1, 0, -1):
                sync.on_ready(layer)
            sync.finish()
        local = torch.arange(32, dtype=torch.float32) * (rank + 1)
        local[4:7] = local[0:3]
        assert torch.equal(store.grad, local)
        for layer in (1, 0, -1):
            sync.on_ready(layer)
        sync.finish()
        want = torch.arange(32, dtype=torch.float32) * 1.5
        want[4:7] = want[0:3]
        assert torch.allclose(store.grad, want)
        out.put((rank, "ok"))
    finally:
        dist.destroy_process_group()


def test_distributed_map_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = sorted(out.get(timeout=5) for _ in range(2))
    assert got == [(0, "ok"), (1, "ok")]


def test_distributed_map_without_process_group():
    assert distributed_map([1, 2, 3], lambda s: [x * 2 for x in s]) == [2, 4, 6]
