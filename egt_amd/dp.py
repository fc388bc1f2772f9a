"""Batch data parallelism for the EGT attention path — one process per GPU.

The reference's only parallelism is tf.distribute.MirroredStrategy
(lib/training/training_base.py:230-247): synchronous single-node DP whose one
collective is the per-step gradient all-reduce.  Here: graphs are independent
units, so the global batch is sharded into contiguous slices per rank (no
data-path collective) and the gradients of all parameters live in ONE flat
contiguous fp32 buffer that is all-reduced once per step over RCCL/xGMI
(torch.distributed backend "nccl" on ROCm; "gloo" in the CPU tests).  The
payload is <= ~2.2 MB, i.e. latency-bound — a single collective, no bucketing.

Two transports for that one collective: ``torch.distributed`` (default) and ``CapiComm`` — the library's own
``egt_dp_init / egt_dp_allreduce / egt_dp_finalize`` (include/egt_amd.h), which put ncclAllReduce
on the stream the backward ran on with nothing of torch.distributed in the step (it is used once, to hand the
RCCL unique id from rank 0 to the other ranks).
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

    def __init__(self, params: Iterable[torch.nn.Parameter], process_group=None, comm: "CapiComm" = None, direct: bool = False):
        """direct: the fused block / FFN backward writes a parameter's gradient straight into its view of the flat buffer and
        hands autograd nothing to accumulate (egt_amd.fused.grad_sinks) — otherwise every parameter costs one elementwise
        `grad += g` launch per step (hundreds of 4-5 us kernels for a 16-layer model).  Valid when every parameter takes part
        in ONE fused call per step (true for the reference's models) and zero() + rebind() run before each backward."""
        self.comm = comm
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
            p._egt_direct_grad = bool(direct)
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
        if self.comm is not None:
            return self.comm.all_reduce_flat(self.flat, average, local_count, global_count)
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


class CapiComm:
    """The per-process RCCL communicator behind the C-ABI (egt_dp_*).  Bootstrap: rank 0 draws the unique id,
    every rank receives it through ``store`` (a torch.distributed Store, default: the default process group's)
    or through ``broadcast`` (callable: bytes-or-None -> bytes), then joins on its current HIP device."""

    KEY = "egt_dp_unique_id"

    def __init__(self, rank: int = None, world: int = None, store=None, broadcast=None):
        import ctypes as C
        from . import _lib
        self._lib = _lib.load()
        self._check = _lib.check
        if rank is None or world is None:
            rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
        buf = C.create_string_buffer(128)
        if rank == 0:
            self._check(self._lib.egt_dp_unique_id(buf))
        ident = bytes(buf.raw)
        if world > 1:
            if broadcast is not None:
                ident = broadcast(ident if rank == 0 else None)
            else:
                if store is None:
                    if not dist.is_initialized():
                        raise RuntimeError("CapiComm: world > 1 needs a store, a broadcast callable or an initialised process group")
                    box = [ident if rank == 0 else None]
                    dist.broadcast_object_list(box, src=0)
                    ident = box[0]
                elif rank == 0:
                    store.set(self.KEY, ident)
                else:
                    ident = bytes(store.get(self.KEY))
        self._check(self._lib.egt_dp_init(C.create_string_buffer(ident, 128), world, rank))
        self.rank, self.world = rank, world

    def all_reduce_flat(self, flat: torch.Tensor, average=True, local_count=None, global_count=None):
        """same contract as dp.all_reduce_flat, on the current stream"""
        if (local_count is None) != (global_count is None):
            raise ValueError("local_count and global_count go together")
        if flat.dtype != torch.float32 or not flat.is_cuda or not flat.is_contiguous():
            raise TypeError("egt_dp_allreduce takes a contiguous fp32 device buffer")
        from ._lib import ptr, current_stream
        weighted = local_count is not None and average
        if weighted:
            flat.mul_(float(local_count) / float(global_count))
        self._check(self._lib.egt_dp_allreduce(ptr(flat), flat.numel(), int(average and not weighted), current_stream()))
        return flat

    def close(self):
        self._check(self._lib.egt_dp_finalize())
