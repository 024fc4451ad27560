"""Multi-GPU use of the layer: one process per GPU, batch sharded, one scalar all-reduce.

Every (clip, frame) is independent in the forward; the only cross-GPU datum of this path is
dL/dlambd, one fp32 scalar.  ``torch.distributed`` with backend "nccl" is RCCL on
ROCm; "gloo" is used by the CPU tests.  No collective touches the forward path.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(global_batch: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous [begin, end) of the clips owned by ``rank``; the first ``global_batch % world`` ranks get one extra."""
    if world < 1 or not (0 <= rank < world) or global_batch < 0:
        raise ValueError("bad shard arguments")
    base, extra = divmod(global_batch, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def allreduce_grad_(grad: torch.Tensor, async_op: bool = False, average: bool = False, group=None):
    """In-place SUM (or mean) all-reduce of a gradient tensor.  With ``async_op`` returns the work handle;
    ``handle.wait()`` orders the caller's current stream after the collective without blocking the host."""
    if not dist.is_available() or not dist.is_initialized():
        raise RuntimeError("torch.distributed is not initialised")
    if average:
        grad.div_(dist.get_world_size(group))
    return dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


def allreduce_lambd_grad(layer, average: bool = True, group=None) -> None:
    """Call after backward(): reduces ``layer.lambd.grad`` over the data-parallel group.  ``average=True``
    matches mean-reduced losses (the reference's losses are, main.py:60,63) as DDP would."""
    g = layer.lambd.grad
    if g is None:
        raise RuntimeError("lambd.grad is None: run backward() first (is the layer trainable?)")
    allreduce_grad_(g, average=average, group=group)
