"""Batch data parallelism for the EGT attention path — one process per GPU.

The reference's only parallelism is tf.distribute.MirroredStrategy
(lib/training/training_base.py:230-247): synchronous single-node DP whose one
collective is the per-step gradient all-reduce.  Here: graphs are independent
units, so the global batch is sharded into contiguous slices per rank (no
data-path collective) and the gradients of all parameters live in ONE flat
contiguous fp32 buffer that is all-reduced once per step over RCCL/xGMI
(torch.distributed backend "nccl" on ROCm; "gloo" in the CPU tests).  The
payload is <= ~2.2 MB, i.e. latency-bound — a single collective, no bucketing.
"""
from __future__ import annotations

from typing import Iterable, List

import torch
import torch.distributed as dist


def shard_batch(n_graphs: int, world_size: int, rank: int):
    """Contiguous slice [lo, hi) of the global batch owned by `rank` (Keras splits
    the global batch across replicas; sizes differ by at most one)."""
    base, rem = divmod(n_graphs, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class FlatGradAllReduce:
    """Owns one flat gradient buffer aliasing every parameter's .grad."""

    def __init__(self, params: Iterable[torch.nn.Parameter], process_group=None):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        n = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(n, device=dev, dtype=dt)
        self.group = process_group
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    @property
    def nbytes(self) -> int:
        return self.flat.numel() * self.flat.element_size()

    def zero(self):
        self.flat.zero_()

    def rebind(self):
        """Re-alias .grad after something replaced it (e.g. zero_grad(set_to_none))."""
        off = 0
        for p in self.params:
            if p.grad is None or p.grad.data_ptr() != self.flat[off:].data_ptr():
                p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def all_reduce(self, average: bool = True, local_count=None, global_count=None, force=False):
        """Sum over ranks, then /world: the loss is a mean over the GLOBAL batch
        (MirroredStrategy semantics).  See all_reduce_flat for uneven shards."""
        return all_reduce_flat(self.flat, self.group, average, local_count, global_count, force)


def flat_grad_view(params, flat) -> bool:
    """True iff `flat` already aliases every parameter's .grad, in order (the layout the
    fused stack backward produces) — then one collective on `flat` reduces everything."""
    if flat is None:
        return False
    off, base = 0, flat.data_ptr()
    for p in params:
        g = p.grad
        if g is None or not g.is_contiguous() or g.data_ptr() != base + off * flat.element_size():
            return False
        off += p.numel()
    return off == flat.numel()


def all_reduce_flat(flat, process_group=None, average=True, local_count=None, global_count=None, force=False):
    """The single per-step collective.

    average=True, counts omitted: SUM over ranks, then /world — the mean over the GLOBAL batch when
    every rank holds a gradient of its LOCAL-mean loss and the shards are equal (what
    MirroredStrategy does with a batch the replica count divides, training_base.py:230-247).
    local_count / global_count given (shard_batch hands out shards that differ by one graph when
    world does not divide the batch): each rank's local-mean gradient is weighted by
    local_count / global_count before the SUM and nothing is divided afterwards, so every graph of
    the global batch carries the same weight.
    average=False: plain SUM (the per-rank loss is already scaled by 1 / global batch).
    force: issue the collective even in a 1-rank group (bench.py / tests: exercises the RCCL call)."""
    if (local_count is None) != (global_count is None):
        raise ValueError("local_count and global_count go together")
    if not (dist.is_available() and dist.is_initialized()):
        return flat
    ws = dist.get_world_size(process_group)
    if ws == 1 and not force:
        return flat
    if local_count is not None and average:
        flat.mul_(float(local_count) / float(global_count))
    if average and local_count is None and dist.get_backend(process_group) == "nccl":
        dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=process_group)   # RCCL averages in the collective: no extra pass over the buffer
        return flat
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=process_group)
    if average and local_count is None:
        flat.div_(ws)
    return flat
