"""Batch sharding across GPUs (one process per GPU).

Sequences are independent, so inference shards the LIST of sequences: rank r works on a
contiguous slice, with a full weight replica, and no collective touches the data path.  The
only communication is the final gather of the (small) results and the barriers used for
timing.  Training adds one exchange step, the gradient all-reduce (see ``allreduce_grads``),
which is what Lightning's DDP strategy does implicitly in the reference (pretraining.py:243).
"""

from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """[begin, end) of rank's contiguous share of n items; sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    q, r = divmod(n, world)
    begin = rank * q + min(rank, r)
    return begin, begin + q + (1 if rank < r else 0)


def distributed_map(items: Sequence, fn: Callable[[Sequence], List], group=None,
                    gather_to: Optional[int] = 0) -> Optional[List]:
    """Apply ``fn`` to this rank's shard of ``items`` and gather the per-item results in the
    original order on rank ``gather_to`` (None: on every rank).  ``fn`` gets the shard as a
    list and must return one result per item.  Works without an initialised process group
    (single process)."""
    if not (dist.is_available() and dist.is_initialized()):
        return list(fn(list(items)))
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    b, e = shard_range(len(items), rank, world)
    mine = list(fn(list(items[b:e])))
    if len(mine) != e - b:
        raise RuntimeError("fn must return one result per item")
    if gather_to is None:
        parts = [None] * world
        dist.all_gather_object(parts, mine, group=group)
    else:
        parts = [None] * world if rank == gather_to else None
        dist.gather_object(mine, parts, dst=gather_to, group=group)
        if rank != gather_to:
            return None
    return [x for part in parts for x in part]


def optimize_codons_sharded(model, proteins, organisms, utrs=None, group=None, **kw):
    """infer_utr.optimize_codons over this rank's shard; full list of RNA strings on rank 0."""
    from .infer_utr import optimize_codons
    idx = list(range(len(proteins)))

    def run(shard):
        if not shard:
            return []
        return optimize_codons(model, [proteins[i] for i in shard], [organisms[i] for i in shard],
                               None if utrs is None else [utrs[i] for i in shard], **kw)

    return distributed_map(idx, run, group=group)


def allreduce_grads(flat_grad: torch.Tensor, group=None) -> torch.Tensor:
    """Mean of a flat gradient buffer over the data-parallel ranks, in place (RCCL all-reduce
    on GPU tensors, gloo on CPU tensors)."""
    if dist.is_available() and dist.is_initialized():
        world = dist.get_world_size(group)
        if world > 1:
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
            flat_grad.div_(world)
    return flat_grad


class GradientSync:
    """Data-parallel gradient mean overlapped with the backward (the role of DDP's bucket hooks
    in the reference, pretraining.py:243).

    Attach to a packed ``train.FlatParameters`` store: ``store.grad_sync = GradientSync(store)``.
    The backward (tc_encoder_backward_notify) reports every layer whose gradients are complete;
    that layer's contiguous range of the flat gradient buffer is then all-reduced on a side
    stream while the kernels of the layers below keep the compute stream busy.  The range holding
    the embeddings, the final LayerNorm and the heads goes last.  ``finish()`` (called by the
    backward) makes the compute stream wait for the exchanges and turns the sums into means.
    Inside ``no_sync()`` nothing is exchanged: micro-batches of a gradient-accumulation step
    only add up locally and the last one, outside the context, exchanges the total."""

    def __init__(self, store, group=None):
        if getattr(store, "layout", None) is None:
            raise ValueError("GradientSync needs a FlatParameters store in the packed layout")
        self.store, self.group = store, group
        self.enabled = True
        self.exchange_single = False      # also run the collectives in a one-rank group (tests)
        starts = [min(o for o, _ in d.values()) for d in store.layout.loff] + [store.layout.total]
        self.layer_ranges = [(starts[i], starts[i + 1]) for i in range(len(store.layout.loff))]
        self.head_range = (0, starts[0])
        self.side = torch.cuda.Stream(device=store.grad.device) if store.grad.is_cuda else None
        self._pending = []
        self._staged = False          # a bucket was exchanged synchronously through the host (gloo)
        self._backend = None          # looked up once: on_ready runs once per layer inside the backward
        self.error = None

    def world(self) -> int:
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def active(self) -> bool:
        if not (dist.is_available() and dist.is_initialized()):
            return False
        return self.enabled and (self.world() > 1 or self.exchange_single)

    class _NoSync:
        def __init__(self, owner):
            self.owner = owner

        def __enter__(self):
            self.prev, self.owner.enabled = self.owner.enabled, False

        def __exit__(self, *exc):
            self.owner.enabled = self.prev

    def no_sync(self):
        return GradientSync._NoSync(self)

    def on_ready(self, layer: int):
        """Callback of the backward: gradients of ``layer`` (or of everything else for -1) are
        complete in stream order."""
        try:
            if layer < 0:
                self.store.sync_shared_bias_grad()
            if not self.active():
                return
            b, e = self.head_range if layer < 0 else self.layer_ranges[layer]
            bucket = self.store.grad[b:e]
            if self._backend is None:
                self._backend = dist.get_backend(self.group)
            if bucket.is_cuda and self._backend == "gloo":
                # gloo (rehearsals on a box with one GPU): staged through host memory, synchronously
                host = bucket.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                bucket.copy_(host)
                self._staged = True
                return
            if bucket.is_cuda:
                ready = torch.cuda.Event()
                ready.record(torch.cuda.current_stream(bucket.device))
                with torch.cuda.stream(self.side):
                    self.side.wait_event(ready)
                    work = dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            else:
                work = dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._pending.append(work)
        except Exception as exc:      # a ctypes callback cannot propagate: keep it for finish()
            self.error = exc

    def abandon(self):
        """After a failed backward: wait for the exchanges already started and forget them; the gradients
        of this step are not usable.  This only leaves the group consistent when every rank failed at the
        same point (a shape or configuration error, which all ranks share).  A failure on ONE rank leaves
        its peers inside all-reduces that this rank never joins: the caller must let the exception end
        the process (the launcher then tears the group down) and not go on to another step."""
        pending, self._pending, self.error = self._pending, [], None
        self._staged = False           # nothing of this step may be divided by the world size later
        for work in pending:
            try:
                work.wait()
            except Exception:      # noqa: BLE001 - the step is lost either way
                pass

    def finish(self):
        """Wait (in stream order) for the exchanges started by on_ready and divide by the world size."""
        err, self.error = self.error, None
        pending, self._pending = self._pending, []
        for work in pending:
            work.wait()
        if err is not None:
            raise err
        if pending or self._staged:
            self.store.grad.div_(self.world())
        self._staged = False
