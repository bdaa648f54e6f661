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


def _flag_device(group):
    return "cuda" if dist.get_backend(group) == "nccl" else "cpu"


class Watchdog:
    """A bounded wait around a call that can hang for ever -- a rank joining a communicator whose other ranks never
    come (ncclCommInitRank, the first collective of a process group).  When `seconds` pass before the block ends, the
    process prints why and EXITS with `exit_code` (os._exit from a timer thread that touches nothing of the GPU): the
    launcher sees a non-zero exit and ends the other ranks (bench.spawn_ranks, torch.distributed.run).  Nothing is
    re-executed and nothing is retried: a process that has initialised the GPU must not be replaced by another program,
    and a half-joined communicator cannot be joined again.  seconds <= 0: no limit."""

    def __init__(self, seconds: float, what: str, exit_code: int = 75):
        self.seconds, self.what, self.exit_code = float(seconds), what, int(exit_code)
        self._timer = None

    def _expire(self):
        import os
        import sys
        sys.stderr.write(f"[pgsd] {self.what}: no progress after {self.seconds:.0f} s -- ending this rank with exit code {self.exit_code}\n")
        sys.stderr.flush()
        os._exit(self.exit_code)

    def __enter__(self):
        if self.seconds > 0:
            import threading
            self._timer = threading.Timer(self.seconds, self._expire)
            self._timer.daemon = True
            self._timer.start()
        return self

    def __exit__(self, *exc):
        if self._timer is not None:
            self._timer.cancel()
        return False


def comm_init_timeout_s() -> float:
    """$PGSD_COMM_INIT_TIMEOUT_S (default 180): the bound on joining a communicator."""
    import os
    try:
        return float(os.environ.get("PGSD_COMM_INIT_TIMEOUT_S", "180"))
    except ValueError:
        return 180.0


def init_library_comm(tree, group: Optional[dist.ProcessGroup] = None) -> bool:
    """Gives `tree` (an SDTree) its own RCCL communicator over the ranks of `group` (pg_comm_init): rank
    0's ncclUniqueId travels through torch.distributed, then every rank joins -- on the calling thread.
    Afterwards tree.allReduce() is the exchange: one ncclAllReduce issued by libpgsd.so itself, the form a
    host without PyTorch would use.

    Whether the library communicator is used at all is decided COLLECTIVELY before anyone joins (one MIN
    all-reduce of a flag, no early return ahead of it).  A rank votes yes when the backend is nccl, every rank
    of this node has a GPU of its own (RCCL refuses two ranks on one device; the check uses the LOCAL world
    size, so multi-node jobs are eligible) AND libpgsd can load RCCL and find its symbols on THIS rank -- every
    rank tries it (ncclGetUniqueId through pg_comm_unique_id; the ids of ranks other than 0 are thrown away), so
    a rank with a broken RCCL installation is known before any rank blocks in ncclCommInitRank (round
    5 checked rank 0 only).  Returns False on every rank when the vote fails -- the caller then exchanges
    through torch.distributed (all_reduce_accumulators), the same RCCL all-reduce issued by PyTorch.
    Once the ranks have agreed to join: a rank on which ncclCommInitRank RETURNS an error (or on which RCCL then
    reports other numbers than the process group's) says so in a second vote and EVERY rank falls back to
    torch.distributed.  A rank that DIES or never arrives before it has joined cannot vote: its peers are blocked
    in ncclCommInitRank and are ended by the Watchdog after comm_init_timeout_s() (exit code 75: the launcher
    ends the job) -- a pre-join failure ends the job, it does not fall back.  The second vote and the letting
    go of a half-agreed communicator run under the same Watchdog.  There is no half-joined communicator and no
    fallback decided by one rank alone."""
    if not dist.is_available() or not dist.is_initialized():
        return False
    import os
    import warnings
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    mine = 1 if (dist.get_backend(group) == "nccl" and torch.cuda.is_available()
                 and local_world <= torch.cuda.device_count()) else 0
    ident = [None]
    if mine:
        try:  # (loads RCCL and resolves its symbols on this rank; only rank 0's id is used)
            my_id = tree.commUniqueId()
            if rank == 0:
                ident[0] = my_id
        except Exception as e:  # noqa: BLE001
            warnings.warn(f"rank {rank}: libpgsd RCCL communicator not available: {e}")
            mine = 0
    flag = torch.tensor([mine], dtype=torch.int32, device=_flag_device(group))
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    if int(flag.item()) != 1:
        return False
    dist.broadcast_object_list(ident, src=0, group=group)
    if ident[0] is None:
        return False
    # ncclCommInitRank blocks until every rank has joined: bounded (see above).  A rank on which the call FAILS -- or on which
    # RCCL then reports other numbers than the process group's (ncclCommCount / ncclCommUserRank read back) -- says so in the
    # second vote: the library communicator is used by every rank or by none (the others let go of theirs).
    ok, why = 1, ""
    with Watchdog(comm_init_timeout_s(), f"rank {rank}: ncclCommInitRank of libpgsd's communicator ({world} ranks), the vote on it"):
        try:
            tree.commInit(world, rank, ident[0])
            n_seen, r_seen = tree.commInfo()
            if (n_seen, r_seen) != (world, rank):
                ok, why = 0, f"RCCL reports {n_seen} ranks / rank {r_seen}, expected {world} / {rank}"
        except Exception as e:  # noqa: BLE001 (whatever the library raises: the vote below decides for everybody)
            ok, why = 0, str(e)
        flag = torch.tensor([ok], dtype=torch.int32, device=_flag_device(group))
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        agreed = int(flag.item()) == 1
        if not agreed:
            if not ok:
                warnings.warn(f"rank {rank}: libpgsd RCCL communicator not usable ({why}); every rank exchanges through torch.distributed")
            try:
                tree.commDestroy()
            except Exception:  # noqa: BLE001
                pass
    if not agreed:
        return False
    return True


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
    spp + s) on EVERY rank: the whole film's samples travel (3 * pixels * spp floats per pass).  Kept for
    shardings that are not bands of rows; the band sharding of the driver uses HaloExchange."""

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


def halo_plan(height: int, stripe_rows: int, rank: int, world: int, reach: int):
    """Which rows of the film travel when every rank develops its own bands (interleaved bands of
    `stripe_rows` rows, band b on rank b % world) with a reconstruction filter that reaches `reach` rows
    beyond a pixel (tent 1, gaussian 2; reach <= stripe_rows, so a band's halo lies in its two neighbour
    bands, which belong to the ring neighbours rank - 1 and rank + 1).  Returns four ascending lists of
    global row numbers: (send_next, send_prev, recv_prev, recv_next) -- the rows this rank owns that
    rank + 1 / rank - 1 need, and the rows it needs from rank - 1 / rank + 1.  send_next of rank r is
    recv_prev of rank r + 1 by construction, so both ends derive the same message."""
    if not (0 < reach <= stripe_rows):
        raise ValueError("the filter must not reach beyond the neighbouring band")
    n_bands = (height + stripe_rows - 1) // stripe_rows

    def needs(q):
        above, below = [], []
        for b in range(q, n_bands, world):
            top = b * stripe_rows
            above += [r for r in range(top - reach, top) if r >= 0]                       # in band b - 1
            below += [r for r in range(top + stripe_rows, top + stripe_rows + reach) if r < height]  # in band b + 1
        return above, below

    recv_prev, recv_next = needs(rank)
    send_next = needs((rank + 1) % world)[0]   # what rank + 1 needs from the bands above its own: mine
    send_prev = needs((rank - 1) % world)[1]
    return send_next, send_prev, recv_prev, recv_next


class HaloExchange:
    """Film development of a band-sharded render without moving the film: every rank develops the pixels
    of its own bands (pg_film_stripes) and fetches only the filter's reach beyond them from its two ring
    neighbours -- per pass and neighbour `reach` rows x width x spp x 12 bytes per band, instead of the
    all-reduce of the whole film's samples LaneGather does.  The developed images of the ranks (zero outside
    their own rows) are summed once per iteration (reduce_image; x + 0 is exact).

    Message order.  A rank issues, in this order: send to next, send to previous, receive from previous, receive
    from next.  With two ranks both neighbours are the same peer, and a backend that ignores tags (NCCL does)
    pairs the messages of one peer in issue order: the peer's FIRST send (its send_next) must be what this rank
    receives FIRST (recv_prev), its second send (send_prev) what it receives second (recv_next).  halo_plan makes
    send_next of rank r equal recv_prev of rank r + 1 row for row, so the pairing holds; _plan() checks that
    equality against the neighbours' own plans once per plan.  (This path has run over gloo only -- with 2, 3
    and 8 ranks, tests/test_multi_rank.py; no multi-GPU hardware is reachable from where it was written.)"""

    def __init__(self, group=None):
        self.group = group
        self._full = None
        self._plans = {}

    def _peer(self, r: int) -> int:
        """dist.P2POp addresses GLOBAL ranks: band-ring index r of `group` -> the global rank."""
        return r if self.group is None else dist.get_global_rank(self.group, r)

    def _plan(self, scene, reach, device):
        key = (scene.film_size, scene.stripe, reach, str(device))
        if key not in self._plans:
            rows, index, count = scene.stripe
            h = scene.film_size[1]
            plan = halo_plan(h, rows, index, count, reach)
            send_next, send_prev, recv_prev, recv_next = plan
            # the same message at both ends (and, for two ranks, the issue order the pairing relies on)
            assert send_next == halo_plan(h, rows, (index + 1) % count, count, reach)[2]
            assert send_prev == halo_plan(h, rows, (index - 1) % count, count, reach)[3]
            own = [y for y in range(h) if (y // rows) % count == index]

            def idx(ys):
                return torch.as_tensor(ys, dtype=torch.long, device=device)

            def local(ys):
                return idx([(y // rows // count) * rows + y % rows for y in ys])

            self._plans[key] = {"rows": plan, "own": idx(own), "n_own": len(own), "send_next": local(send_next),
                                "send_prev": local(send_prev), "recv_prev": idx(recv_prev), "recv_next": idx(recv_next)}
        return self._plans[key]

    def __call__(self, L: torch.Tensor, scene, spp: int, reach: int) -> torch.Tensor:
        """(3, pixels * spp) in full-frame lane order, valid on this rank's rows and `reach` rows around its bands."""
        if scene.stripe is None:
            raise ValueError("HaloExchange needs the band sharding of WavefrontScene.set_shard(rank, world, stripe_rows > 0)")
        w, h = scene.film_size
        rows, index, count = scene.stripe
        n = 3 * w * h * spp
        if self._full is None or self._full.numel() != n or self._full.device != L.device:
            self._full = torch.zeros((3, h, w * spp), dtype=L.dtype, device=L.device)
        full = self._full
        pl = self._plan(scene, reach, L.device)
        Lr = L.reshape(3, pl["n_own"], w * spp)
        full[:, pl["own"]] = Lr
        nxt, prv = self._peer((index + 1) % count), self._peer((index - 1) % count)
        host = dist.get_backend(self.group) == "gloo" and L.is_cuda  # (gloo moves host memory)

        def pick(ix):
            t = Lr[:, ix].contiguous()
            return t.cpu() if host else t

        def room(ix):
            return torch.empty((3, ix.numel(), w * spp), dtype=L.dtype, device="cpu" if host else L.device)

        ops, got = [], []
        # (tags keep the two messages of one peer apart where the backend honours them; where it does not, the issue order
        # documented above does)
        if pl["send_next"].numel():
            ops.append(dist.P2POp(dist.isend, pick(pl["send_next"]), nxt, self.group, 0))
        if pl["send_prev"].numel():
            ops.append(dist.P2POp(dist.isend, pick(pl["send_prev"]), prv, self.group, 1))
        if pl["recv_prev"].numel():
            got.append((pl["recv_prev"], room(pl["recv_prev"])))
            ops.append(dist.P2POp(dist.irecv, got[-1][1], prv, self.group, 0))
        if pl["recv_next"].numel():
            got.append((pl["recv_next"], room(pl["recv_next"])))
            ops.append(dist.P2POp(dist.irecv, got[-1][1], nxt, self.group, 1))
        self.bytes_last_pass = sum(op.tensor.numel() * 4 for op in ops if op.op is dist.isend)
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        for ix, buf in got:
            full[:, ix] = buf.to(L.device)
        return full.reshape(3, h * w * spp)

    def reduce_image(self, image: torch.Tensor) -> torch.Tensor:
        """The film of all ranks from the images they developed for their own rows (zero elsewhere)."""
        return _all_reduce_sum(image.clone(), self.group)


def lowest_rank_with(flag: bool, group: Optional[dist.ProcessGroup] = None) -> int:
    """The lowest rank of `group` whose `flag` is set, or -1 when none is -- the same number on every rank (one MIN
    all-reduce): how the ranks of a sharded render agree on who writes files, so that no collective depends on a
    per-rank argument."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0 if flag else -1
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    t = torch.tensor([rank if flag else world], dtype=torch.int64, device=_flag_device(group))
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    r = int(t.item())
    return r if r < world else -1


def min_max_over_ranks(value: float, group: Optional[dist.ProcessGroup] = None):
    """(min, max) of a per-rank number over the ranks."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return value, value
    t = torch.tensor([-float(value), float(value)], dtype=torch.float64, device=_flag_device(group))
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return -float(t[0].item()), float(t[1].item())


def max_over_ranks(seconds: float, device=None, group: Optional[dist.ProcessGroup] = None) -> float:
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return seconds
    backend = dist.get_backend(group)
    t = torch.tensor([seconds], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
