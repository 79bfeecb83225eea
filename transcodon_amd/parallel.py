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
