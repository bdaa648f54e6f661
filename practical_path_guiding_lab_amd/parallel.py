"""Multi-GPU plumbing for the SD-tree path: one process per GPU, rays sharded across ranks, one
exact int64 all-reduce of the accumulators per training iteration.

The reference has no multi-GPU code; this is new design around
refineAndPrepareSDTreeForNextIteration (path_guiding_integrator.py:566-586).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous share [start, start+count) of n rays/pixels for `rank`; shares differ by <= 1."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


def all_reduce_accumulators(acc: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """Element-wise sum of the int64 accumulator buffer over all ranks, in place.

    With the `nccl` backend this is one RCCL all-reduce over xGMI on the tensor that aliases
    library memory (pg_accumulators).  With `gloo` (CPU tests, or ranks sharing one GPU) CUDA
    tensors are staged through the host.  Limbs carry 32 payload bits in 64, so sums over any
    realistic number of ranks cannot overflow and no carry handling is needed."""
    if acc.dtype != torch.int64:
        raise TypeError("accumulators must be int64")
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return acc
    if acc.numel() == 0:
        return acc
    backend = dist.get_backend(group)
    if acc.is_cuda and backend == "gloo":
        host = acc.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        acc.copy_(host)
    else:
        dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
    return acc


def max_over_ranks(seconds: float, device=None, group: Optional[dist.ProcessGroup] = None) -> float:
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return seconds
    backend = dist.get_backend(group)
    t = torch.tensor([seconds], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
