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


def init_library_comm(tree, group: Optional[dist.ProcessGroup] = None, timeout_s: float = 120.0) -> bool:
    """Gives `tree` (an SDTree) its own RCCL communicator over the ranks of `group` (pg_comm_init): rank
    0's ncclUniqueId travels through torch.distributed, then every rank joins.  Afterwards
    tree.allReduce() is the exchange -- one ncclAllReduce issued by libpgsd.so itself, the form a host
    without PyTorch would use.  Returns False on EVERY rank (and leaves no communicator behind) when the
    ranks do not each have a GPU of their own (RCCL refuses two ranks on one device), RCCL cannot be
    loaded, or any rank fails to join within `timeout_s`: the caller then exchanges through
    torch.distributed (all_reduce_accumulators), which is the same RCCL all-reduce issued by PyTorch."""
    if not dist.is_available() or not dist.is_initialized():
        return False
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if world > torch.cuda.device_count():
        return False
    ident = [None]
    err = None
    if rank == 0:
        try:
            ident[0] = tree.commUniqueId()
        except Exception as e:  # RCCL missing: tell the others instead of leaving them waiting
            err = e
    dist.broadcast_object_list(ident, src=0, group=group)
    if ident[0] is None:
        if err is not None and rank == 0:
            import warnings
            warnings.warn(f"libpgsd RCCL communicator not available: {err}")
        return False
    # join on a helper thread (the call leaves the interpreter lock): a rank that cannot join must not
    # leave the others waiting in ncclCommInitRank for ever
    import threading
    outcome = {}

    def join():
        try:
            tree.commInit(world, rank, ident[0])
            outcome["ok"] = True
        except Exception as e:
            outcome["err"] = e

    th = threading.Thread(target=join, daemon=True)
    th.start()
    th.join(timeout_s)
    mine = 1 if outcome.get("ok") else 0
    flag = torch.tensor([mine], dtype=torch.int32, device="cuda" if dist.get_backend(group) == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    if int(flag.item()) == 1:
        return True
    import warnings
    warnings.warn(f"libpgsd RCCL communicator: rank {rank} {'joined' if mine else 'did not join'}"
                  f"{' (' + str(outcome['err']) + ')' if 'err' in outcome else ''}; falling back to torch.distributed")
    if mine:
        tree.commDestroy()
    return False


def _all_reduce_sum(t: torch.Tensor, group=None) -> torch.Tensor:
    """In-place sum over ranks of any tensor; CUDA tensors go through the host under gloo."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1 or t.numel() == 0:
        return t
    if t.is_cuda and dist.get_backend(group) == "gloo":
        host = t.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        t.copy_(host)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def all_reduce_sums(sumL: torch.Tensor, sumL2: torch.Tensor, group=None):
    """Whole-film copies of the per-pixel sums of a tile-sharded render: every rank holds its own
    pixels and zeros elsewhere (tiles are disjoint), so the sum over ranks is exact and identical on
    every rank -- variance, MSE and the stop-training decision (main.py:334-377) then agree."""
    return _all_reduce_sum(sumL.clone(), group), _all_reduce_sum(sumL2.clone(), group)


class LaneGather:
    """Collects the lanes every rank traced for its tile into full-frame lane order (lane = pixel *
    spp + s), for the film reconstruction that needs a pixel's neighbours (render.render)."""

    def __init__(self, group=None):
        self.group = group

    def __call__(self, L: torch.Tensor, scene, spp: int) -> torch.Tensor:
        w, h = scene.film_size
        npix = w * h
        pix = torch.from_numpy(scene.local_pixels()).to(L.device)
        lanes = (pix[:, None] * spp + torch.arange(spp, device=L.device)[None, :]).reshape(-1)
        full = torch.zeros((3, npix * spp), dtype=L.dtype, device=L.device)
        full[:, lanes] = L
        return _all_reduce_sum(full, self.group)  # disjoint tiles: x + 0 is exact


def max_over_ranks(seconds: float, device=None, group: Optional[dist.ProcessGroup] = None) -> float:
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return seconds
    backend = dist.get_backend(group)
    t = torch.tensor([seconds], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
