"""N>1 path under world_size 2.

CPU (gloo): the exchange step -- per-rank accumulators in the device's limb format, summed with
practical_path_guiding_lab_amd.parallel.all_reduce_accumulators, resolve to exactly the
accumulators of a single process that saw all records (integer sums are order- and
partition-independent).  The per-rank accumulators come from the oracle (no GPU here).

GPU (-m gpu): two ranks sharing cuda:0 over gloo run the real product path (splat shard ->
all-reduce -> refine) and must end with the single-rank tree, bit for bit.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import synth
from oracle import pg_oracle as po

BB0, BB1 = [0.0] * 3, [100.0] * 3
M = 60_000


def _limbs(lo: np.ndarray, hi: np.ndarray) -> np.ndarray:
    """128-bit two's complement (lo, hi) -> three int64 limbs with value l0 + l1*2^32 + l2*2^64."""
    l0 = (lo & np.uint64(0xFFFFFFFF)).astype(np.int64)
    l1 = (lo >> np.uint64(32)).astype(np.int64)
    return np.stack([l0, l1, hi.astype(np.int64)], axis=1)


def _resolve(limbs: np.ndarray):
    return [int(a) + (int(b) << 32) + (int(c) << 64) for a, b, c in limbs]


def _base_tree():
    pair = synth.build_skewed(1 << 13, 3)
    return pair.prev.export()


def _rank_records(rank, world):
    from practical_path_guiding_lab_amd.parallel import shard

    rec = synth.records(M, 4242, BB0, BB1)
    s, c = shard(M, rank, world)
    return {k: np.ascontiguousarray(v[..., s:s + c]) for k, v in rec.items()}


def _worker_cpu(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from practical_path_guiding_lab_amd.parallel import all_reduce_accumulators, max_over_ranks

    t = po.OracleTree()
    t.load(_base_tree())
    t.reset()
    synth.splat(t, _rank_records(rank, world))
    limbs = _limbs(t.quad_column("acc_lo"), t.quad_column("acc_hi"))
    cnt = t.kd_column("count").astype(np.int64)
    buf = torch.from_numpy(np.concatenate([limbs.reshape(-1), cnt]))
    all_reduce_accumulators(buf)
    assert max_over_ranks(float(rank)) == float(world - 1)
    if rank == 0:
        np.save(out, buf.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_covers_everything():
    from practical_path_guiding_lab_amd.parallel import shard

    for n in (0, 1, 7, 8, 1000003):
        for w in (1, 2, 3, 8):
            parts = [shard(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and sum(c for _, c in parts) == n
            for (s0, c0), (s1, _) in zip(parts, parts[1:]):
                assert s0 + c0 == s1
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1
    with pytest.raises(ValueError):
        shard(10, 2, 2)


@pytest.mark.parametrize("world,port", [(2, 29611), (8, 29624)])
def test_rank_allreduce_equals_single_process(tmp_path, world, port):
    """2 and 8 ranks (the node the multi-GPU configurations name): every rank splats its share of the records, the
    accumulators are summed over gloo in the device's limb format, and equal a single process's that saw all records."""
    out = str(tmp_path / "sum.npy")
    mp.spawn(_worker_cpu, args=(world, port, out), nprocs=world, join=True)
    got = np.load(out)
    ref = po.OracleTree()
    ref.load(_base_tree())
    ref.reset()
    synth.splat(ref, synth.records(M, 4242, BB0, BB1))
    nq = ref.quad_size
    exp = [(int(h) << 64) + int(l) for l, h in zip(ref.quad_column("acc_lo"), ref.quad_column("acc_hi"))]
    # two's complement of negative totals: compare modulo 2^128 like the hardware does
    got_v = [v % (1 << 128) for v in _resolve(got[: nq * 3].reshape(nq, 3))]
    assert got_v == [v % (1 << 128) for v in exp]
    np.testing.assert_array_equal(got[nq * 3:], ref.kd_column("count").astype(np.int64))


def _worker_cpu_packed(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from practical_path_guiding_lab_amd import _native as N
    from practical_path_guiding_lab_amd.parallel import all_reduce_accumulators

    t = po.OracleTree()
    t.load(_base_tree())
    t.reset()
    rec = _rank_records(rank, world)
    rec["radiance"][::5] *= np.float32(-3.0)        # negative totals: the signed top word
    rec["radiance"][1::9] = np.float32(3e38)         # clamped to 2^48: the biggest q a record can carry
    synth.splat(t, rec)
    limbs = _limbs(t.quad_column("acc_lo"), t.quad_column("acc_hi"))
    nq = limbs.shape[0]
    # the device's four-word accumulator: three limbs + a count (here: this rank's number of records, the same for every node)
    acc = np.concatenate([limbs, np.full((nq, 1), rec["radiance"].shape[0], np.int64)], axis=1)
    # limbs as the device leaves them after many adds: payloads beyond 32 bits, mixed signs, the same value
    acc[:, 0] += np.int64(5) << np.int64(32)
    acc[:, 1] -= np.int64(5)
    acc[:, 1] += np.int64(-7) << np.int64(32)
    acc[:, 2] -= np.int64(-7)
    packed = np.zeros((nq, 3), np.int64)
    L = N.lib()
    assert L.pg_exchange_pack_words(acc.ctypes.data, nq, packed.ctypes.data) == 0
    assert packed[:, :2].min() >= 0 and packed[:, :2].max() < (1 << 52)
    buf = torch.from_numpy(packed.reshape(-1).copy())
    all_reduce_accumulators(buf)
    back = np.zeros((nq, 4), np.int64)
    assert L.pg_exchange_unpack_words(buf.numpy().ctypes.data, nq, back.ctypes.data) == 0
    if rank == 0:
        np.save(out, back)
    dist.barrier()
    dist.destroy_process_group()


def test_packed_exchange_format_with_eight_ranks(tmp_path):
    """The 24-byte exchange format (pgsd.h pg_exchange_pack: two 52-bit pieces + count << 24 | top) through the LIBRARY's
    own arithmetic on host arrays (pg_exchange_pack_words / _unpack_words: the functions the device kernels inline): eight
    ranks pack their accumulators, gloo sums the packed words, the unpacked result is the single-process sum -- value and
    count of every node -- with negative totals, clamped 2^88 weights and limbs that are not normalised."""
    world = 8
    out = str(tmp_path / "packed.npy")
    mp.spawn(_worker_cpu_packed, args=(world, 29627, out), nprocs=world, join=True)
    got = np.load(out)
    ref = po.OracleTree()
    ref.load(_base_tree())
    ref.reset()
    rec = synth.records(M, 4242, BB0, BB1)
    from practical_path_guiding_lab_amd.parallel import shard
    for r in range(world):   # (the per-rank edits, applied to the same records)
        s0, c0 = shard(M, r, world)
        rec["radiance"][s0:s0 + c0][::5] *= np.float32(-3.0)
        rec["radiance"][s0:s0 + c0][1::9] = np.float32(3e38)
    synth.splat(ref, rec)
    exp = [((int(h) << 64) + int(l)) % (1 << 128) for l, h in zip(ref.quad_column("acc_lo"), ref.quad_column("acc_hi"))]
    assert [v % (1 << 128) for v in _resolve(got[:, :3])] == exp
    assert (got[:, 3] == M).all()
    assert (got[:, 0] >= 0).all() and (got[:, 0] < (1 << 32)).all() and (got[:, 1] >= 0).all() and (got[:, 1] < (1 << 32)).all()
    assert any(v >= (1 << 127) for v in exp) and max(abs(int(x)) for x in got[:, 2]) > 0   # negative totals, high limbs in use


def test_packed_exchange_headroom_at_the_bounds():
    """The bounds the format states: 2^31 records of |q| = 2^88 per accumulator over all ranks, any split over up to 2048
    ranks -- the packed words of the ranks' extreme shares add without overflow and decode exactly."""
    from practical_path_guiding_lab_amd import _native as N

    L = N.lib()
    for ranks in (1, 8, 2048):
        per = (1 << 31) // ranks                       # records per rank
        for sign in (1, -1):
            T = sign * per * (1 << 88)                 # this rank's total
            Tm = T % (1 << 128)
            l0, l1 = Tm & 0xFFFFFFFF, (Tm >> 32) & 0xFFFFFFFF
            l2 = (Tm >> 64) - (1 << 64) if (Tm >> 127) else (Tm >> 64)
            acc = np.array([[l0, l1, l2, per]], np.int64)
            p = np.zeros((1, 3), np.int64)
            assert L.pg_exchange_pack_words(acc.ctypes.data, 1, p.ctypes.data) == 0
            tot = [int(x) * ranks for x in p[0]]       # every rank the same share: the element-wise sums
            assert all(-(1 << 63) <= x < (1 << 63) for x in tot)
            q = np.array([tot], np.int64)
            back = np.zeros((1, 4), np.int64)
            assert L.pg_exchange_unpack_words(q.ctypes.data, 1, back.ctypes.data) == 0
            assert _resolve(back[:, :3])[0] == T * ranks and int(back[0, 3]) == per * ranks


def test_watchdog_ends_a_rank_that_waits_for_ever(tmp_path):
    """parallel.Watchdog, the bound around ncclCommInitRank and the first barrier: a block that outlives its limit ends the
    PROCESS with the given exit code (the launcher then ends the other ranks); a block that finishes in time is untouched."""
    import subprocess
    import time as _t
    code = ("import sys, time; sys.path.insert(0, %r)\n"
            "from practical_path_guiding_lab_amd.parallel import Watchdog\n"
            "with Watchdog(30, 'quick'):\n    pass\n"
            "with Watchdog(0.5, 'a rank that never comes', exit_code=75):\n    time.sleep(60)\n"
            "print('not reached')\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    t0 = _t.time()
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 75 and "not reached" not in r.stdout and "a rank that never comes" in r.stderr
    assert _t.time() - t0 < 45


def test_single_rank_is_a_noop():
    from practical_path_guiding_lab_amd.parallel import all_reduce_accumulators

    a = torch.arange(10, dtype=torch.int64)
    assert all_reduce_accumulators(a) is a
    with pytest.raises(TypeError):
        all_reduce_accumulators(torch.zeros(3, dtype=torch.int32))


# ---------------------------------------------------------------------------------------------
def _worker_gpu(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from practical_path_guiding_lab_amd.parallel import all_reduce_accumulators
    from practical_path_guiding_lab_amd.sdtree import SDTree

    g = SDTree(0)
    g.load(_base_tree())
    g.setIteration(3, False)
    rec = _rank_records(rank, world)
    g.addDataPropagate({k: torch.from_numpy(v).cuda() for k, v in rec.items()})
    # the exchange format: pack (24 B per accumulator) -> sum -> unpack, and beside it the raw 32-byte layout of the same
    # per-rank accumulators summed the old way: the same value and count for every accumulator
    raw = g.accumulators().clone()
    all_reduce_accumulators(g.packAccumulators())
    g.unpackAccumulators()
    kd_p, lo_p, hi_p = g.exportAccumulators()
    all_reduce_accumulators(raw)
    g.accumulators().copy_(raw)
    kd_r, lo_r, hi_r = g.exportAccumulators()
    assert (kd_p == kd_r).all() and (lo_p == lo_r).all() and (hi_p == hi_r).all() and int(kd_p[0]) == M
    g.refineAndPrepare()
    e = g.export()
    np.savez(out % rank, **e)
    dist.barrier()
    dist.destroy_process_group()


def _worker_render(rank, world, port, out, mode):
    """mode 'tiles': ranks split the film rows; mode 'passes': ranks take alternate passes."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from practical_path_guiding_lab_amd.integrator import PathGuidingIntegrator
    from practical_path_guiding_lab_amd.parallel import all_reduce_accumulators, shard
    from practical_path_guiding_lab_amd.render import IndependentSampler, WavefrontScene
    from practical_path_guiding_lab_amd.scene import cornell_box

    sc = cornell_box(RES, RES, 6, 8)
    g = PathGuidingIntegrator({"max_depth": 6, "rr_depth": 8})
    g.setup(RES * RES, sc.bbox_min - np.float32(1e-4), sc.bbox_max + np.float32(1e-4), 20, 20, True, 0.5)
    ws = WavefrontScene(sc)
    if mode == "tiles":
        ws.pixel_range = shard(RES * RES, rank, world)
    Ls = []
    cumm = 0
    for k in range(3):
        g.setIteration(k, False)
        for p in range(2 ** (k + 2)):
            if mode == "tiles" or p % world == rank:
                L, _, _ = g.sample(ws, IndependentSampler(1, 77 + cumm + p))
                if k == 2 and p < 2:
                    Ls.append(L.cpu().numpy())
        cumm += 2 ** (k + 2)
        g.refineAndPrepareSDTreeForNextIteration(all_reduce_accumulators)
    sums = g.sumL.cpu()
    dist.all_reduce(sums)
    np.savez(out % rank, sumL=sums.numpy(), L=np.stack(Ls) if Ls else np.zeros(0), **g.sdTree.export())
    dist.barrier()
    dist.destroy_process_group()


RES = 24


@pytest.mark.gpu
@pytest.mark.parametrize("mode,port", [("tiles", 29613), ("passes", 29614)])
def test_two_ranks_render_like_one(tmp_path, mode, port):
    from practical_path_guiding_lab_amd.parallel import shard
    from practical_path_guiding_lab_amd.scene import cornell_box

    world = 2
    out = str(tmp_path / "r%d.npz")
    mp.spawn(_worker_render, args=(world, port, out, mode), nprocs=world, join=True)
    sc = cornell_box(RES, RES, 6, 8)
    o = po.OracleSDTreePair()
    o.setup(sc.bbox_min - np.float32(1e-4), sc.bbox_max + np.float32(1e-4), 20, 20, True)
    sumL = np.zeros((3, RES * RES), np.float32)
    sumL2 = np.zeros_like(sumL)
    cumm, keepL = 0, []
    for k in range(3):
        for p in range(2 ** (k + 2)):
            L, _ = po.render_pass(o, sc, sc.camera, 6, 8, k, False, 77 + cumm + p, 1, True, 0.5, sumL, sumL2)
            if k == 2 and p < 2:
                keepL.append(L)
        cumm += 2 ** (k + 2)
        o.refine_and_prepare(k)
    exp = o.prev.export()
    for r in range(world):
        got = dict(np.load(out % r))
        for key in exp:  # every rank ends with the single-process tree, bit for bit
            np.testing.assert_array_equal(np.asarray(got[key]).astype(np.float64), np.asarray(exp[key]).astype(np.float64), err_msg=key)
        if mode == "tiles":  # a tile's lanes are the corresponding slice of the full-frame pass
            b, c = shard(RES * RES, r, world)
            for i in range(2):
                np.testing.assert_array_equal(got["L"][i].view(np.uint32), keepL[i][:, b:b + c].view(np.uint32))
            # disjoint tiles: the summed per-pixel sums are exactly the full-frame sums
            np.testing.assert_array_equal(got["sumL"].view(np.uint32), sumL.view(np.uint32))
        else:  # passes interleaved over ranks: same samples, fp32 sums in a different order
            np.testing.assert_allclose(got["sumL"], sumL, rtol=2e-5, atol=1e-6)


@pytest.mark.gpu
def test_two_ranks_on_gpu_match_single_rank_and_oracle(tmp_path):
    world = 2
    out = str(tmp_path / "tree%d.npz")
    mp.spawn(_worker_gpu, args=(world, 29612, out), nprocs=world, join=True)
    o = po.OracleSDTreePair()
    o.current.load(_base_tree())
    o.current.reset()
    synth.splat(o.current, synth.records(M, 4242, BB0, BB1))
    o.refine_and_prepare(3)
    exp = o.prev.export()
    for r in range(world):
        got = dict(np.load(out % r))
        assert set(got) == set(exp)
        for k in exp:
            np.testing.assert_array_equal(np.asarray(got[k]).astype(np.float64), np.asarray(exp[k]).astype(np.float64), err_msg=k)


# ---------------------------------------------------------------------------------------------
# tile-sharded end-to-end render: the driver of main.py with two ranks equals the driver with one
def test_stripes_partition_the_film():
    """set_shard: interleaved bands (pg_pass_params stripes) or one contiguous range per rank; the
    ranks' pixels are disjoint and cover the film, for sizes that do not divide evenly too."""
    from practical_path_guiding_lab_amd.render import WavefrontScene
    from practical_path_guiding_lab_amd.scene import cornell_box

    for (w, h), world, rows in (((24, 18), 2, 4), ((16, 9), 3, 2), ((8, 5), 8, 1), ((12, 7), 3, 0)):
        seen = np.zeros(w * h, int)
        for r in range(world):
            ws = WavefrontScene(cornell_box(w, h, 4, 8))
            ws.set_shard(r, world, rows)
            assert ws.sharded
            px = ws.local_pixels()
            assert (np.diff(px) > 0).all()
            seen[px] += 1
            if rows:  # band b of `rows` rows belongs to rank b % world
                assert ((px // w // rows) % world == r).all()
        assert (seen == 1).all()
    ws = WavefrontScene(cornell_box(8, 8, 4, 8))
    ws.set_shard(0, 1)
    assert not ws.sharded and ws.local_pixels().shape[0] == 64


def _worker_gather(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from practical_path_guiding_lab_amd.parallel import LaneGather, all_reduce_sums
    from practical_path_guiding_lab_amd.render import WavefrontScene
    from practical_path_guiding_lab_amd.scene import cornell_box

    w, h, spp = 10, 7, 3
    ws = WavefrontScene(cornell_box(w, h, 4, 8))
    ws.set_shard(rank, world, 2)
    px = ws.local_pixels()
    lanes = (px[:, None] * spp + np.arange(spp)[None, :]).reshape(-1)
    full = np.arange(3 * w * h * spp, dtype=np.float32).reshape(3, -1) * 0.5 + 1.0  # what a single rank would have traced
    got = LaneGather()(torch.from_numpy(full[:, lanes].copy()), ws, spp)
    sumL = torch.zeros(3, w * h)
    sumL[:, torch.from_numpy(px)] = float(rank + 1)
    s1, s2 = all_reduce_sums(sumL, sumL * 2)
    owner = np.zeros(w * h)
    for r in range(world):
        ws.set_shard(r, world, 2)
        owner[ws.local_pixels()] = r + 1
    ok = bool((got.numpy() == full).all() and (s1.numpy() == owner[None, :]).all() and (s2.numpy() == 2 * owner[None, :]).all()
              and float(sumL.sum()) == 3.0 * (rank + 1) * px.shape[0])  # the rank's own arrays are untouched
    np.save(out % rank, np.array([ok]))
    dist.barrier()
    dist.destroy_process_group()


def test_lane_gather_and_reduced_sums_two_ranks(tmp_path):
    """parallel.LaneGather puts every rank's tile lanes into full-frame lane order; all_reduce_sums
    gives every rank the whole film's per-pixel sums (gloo, CPU tensors, world 2)."""
    out = str(tmp_path / "g%d.npy")
    mp.spawn(_worker_gather, args=(2, 29615, out), nprocs=2, join=True)
    assert all(bool(np.load(out % r)[0]) for r in range(2))


def test_halo_plan_is_the_same_message_at_both_ends():
    """parallel.halo_plan: what rank r sends down is what rank r + 1 expects from above (and the other way round),
    every row a band's filter reaches is covered, nothing but `reach` rows per band and neighbour travels."""
    from practical_path_guiding_lab_amd.parallel import halo_plan
    for h, rows, world, reach in ((22, 4, 2, 1), (22, 4, 3, 2), (1080, 4, 8, 1), (9, 2, 5, 2), (16, 4, 4, 2), (7, 4, 2, 1)):
        plans = [halo_plan(h, rows, r, world, reach) for r in range(world)]
        for r in range(world):
            sn, sp, rp, rn = plans[r]
            assert sn == plans[(r + 1) % world][2] and sp == plans[(r - 1) % world][3]
            own = {y for y in range(h) if (y // rows) % world == r}
            assert set(sn) <= own and set(sp) <= own and not (set(rp) | set(rn)) & own
            need = {y + d for y in own for d in range(-reach, reach + 1) if 0 <= y + d < h} - own
            assert need == set(rp) | set(rn)
            n_bands = len(range(r, (h + rows - 1) // rows, world))
            assert len(sn) <= reach * n_bands + reach and len(sp) <= reach * n_bands + reach
    with pytest.raises(ValueError):
        halo_plan(16, 1, 0, 2, 2)  # the gaussian filter reaches beyond a one-row band's neighbour


def _worker_halo(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from practical_path_guiding_lab_amd.parallel import HaloExchange
    from practical_path_guiding_lab_amd.render import WavefrontScene
    from practical_path_guiding_lab_amd.scene import cornell_box

    ok = True
    for (w, h, spp, rows, reach) in ((10, 22, 3, 4, 1), (6, 17, 2, 4, 2), (5, 9, 1, 2, 2)):
        ws = WavefrontScene(cornell_box(w, h, 4, 8))
        ws.set_shard(rank, world, rows)
        px = ws.local_pixels()
        lanes = (px[:, None] * spp + np.arange(spp)[None, :]).reshape(-1)
        full = np.arange(3 * w * h * spp, dtype=np.float32).reshape(3, -1) * 0.5 + 1.0  # what a single rank would have traced
        hx = HaloExchange()
        got = hx(torch.from_numpy(full[:, lanes].copy()), ws, spp, reach).numpy().reshape(3, h, w * spp)
        own = sorted({int(p) // w for p in px})
        valid = sorted({y + d for y in own for d in range(-reach, reach + 1) if 0 <= y + d < h})
        ok = ok and bool((got[:, valid] == full.reshape(3, h, w * spp)[:, valid]).all())
        # per pass and neighbour at most `reach` rows per band: 3 channels x rows x width x spp x 4 bytes
        n_bands = len({y // rows for y in own})
        ok = ok and hx.bytes_last_pass <= 2 * (n_bands + 1) * reach * w * spp * 12
        ok = ok and hx.bytes_last_pass < 3 * w * h * spp * 4  # (less than the film LaneGather would move)
        img = torch.zeros(3, h * w)
        img[:, torch.from_numpy(px)] = float(rank + 1)
        tot = hx.reduce_image(img)
        ok = ok and bool((tot.min() >= 1).item() and float(img.sum()) == 3.0 * (rank + 1) * px.shape[0])
    np.save(out % rank, np.array([ok]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,port", [(2, 29617), (3, 29618), (8, 29621)])
def test_halo_exchange_moves_the_filters_reach_not_the_film(tmp_path, world, port):
    """parallel.HaloExchange over gloo (CPU tensors): after the exchange a rank holds its own rows and the rows its
    reconstruction filter reaches (one for the tent filter, two for the gaussian) exactly as a single rank traced
    them, having sent no more than that many rows per band to each ring neighbour -- with two ranks both neighbours
    are the same process and the two messages are kept apart by their tags."""
    out = str(tmp_path / "h%d.npy")
    mp.spawn(_worker_halo, args=(world, port, out), nprocs=world, join=True)
    assert all(bool(np.load(out % r)[0]) for r in range(world))


def _worker_subgroup(rank, world, port, out):
    """Three processes, the band ring is the SUB-group of global ranks 1 and 2 (ring index 0 = global rank 1): dist.P2POp
    addresses global ranks, so a ring index handed to it unchanged would talk to the wrong process."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from practical_path_guiding_lab_amd.parallel import HaloExchange, lowest_rank_with
    from practical_path_guiding_lab_amd.render import WavefrontScene
    from practical_path_guiding_lab_amd.scene import cornell_box

    ring = dist.new_group([1, 2])
    ok = True
    # who writes files: the lowest rank that was given a directory, the same answer everywhere, -1 when nobody was
    ok = ok and lowest_rank_with(rank == 2) == 2 and lowest_rank_with(rank >= 1) == 1 and lowest_rank_with(False) == -1
    ok = ok and lowest_rank_with(True) == 0
    if rank >= 1:
        idx = rank - 1
        ok = ok and lowest_rank_with(idx == 1, ring) == 1 and lowest_rank_with(False, ring) == -1
        w, h, spp, rows, reach = 6, 13, 2, 2, 1
        ws = WavefrontScene(cornell_box(w, h, 4, 8))
        ws.set_shard(idx, 2, rows)
        px = ws.local_pixels()
        lanes = (px[:, None] * spp + np.arange(spp)[None, :]).reshape(-1)
        full = np.arange(3 * w * h * spp, dtype=np.float32).reshape(3, -1) + 0.25
        hx = HaloExchange(ring)
        for _ in range(2):  # (the second pass runs on the cached plan)
            got = hx(torch.from_numpy(full[:, lanes].copy()), ws, spp, reach).numpy().reshape(3, h, w * spp)
            own = sorted({int(p) // w for p in px})
            valid = sorted({y + d for y in own for d in range(-reach, reach + 1) if 0 <= y + d < h})
            ok = ok and bool((got[:, valid] == full.reshape(3, h, w * spp)[:, valid]).all())
        ok = ok and len(hx._plans) == 1
    np.save(out % rank, np.array([ok]))
    dist.barrier()
    dist.destroy_process_group()


def test_halo_exchange_in_a_sub_group_and_the_writer_vote(tmp_path):
    out = str(tmp_path / "s%d.npy")
    mp.spawn(_worker_subgroup, args=(3, 29622, out), nprocs=3, join=True)
    assert all(bool(np.load(out % r)[0]) for r in range(3))


def _drive(shard_arg, out_dir=None, spp_per_pass=2, **kw):
    from practical_path_guiding_lab_amd.driver import run_guided_render
    from practical_path_guiding_lab_amd.integrator import PathGuidingIntegrator
    from practical_path_guiding_lab_amd.render import WavefrontScene
    from practical_path_guiding_lab_amd.scene import cornell_box

    sc = cornell_box(RES, 20, 6, 8)  # tent filter: a pixel needs its neighbours, which another rank traced
    g = PathGuidingIntegrator({"max_depth": 6, "rr_depth": 8})
    gt = torch.full((3, RES * 20), 0.25, device="cuda")
    res = run_guided_render(WavefrontScene(sc), g, budget_spp=60, initial_seed=5, ground_truth=gt, training_spp_per_pass=spp_per_pass,
                            batch_spp=4, log=lambda s: None, shard=shard_arg, out_dir=out_dir, **kw)
    rows = {k: np.array(v.rows, dtype=np.float64)[:, 1:] for k, v in res["records"].items() if v.rows}  # (all but the wall time)
    return res["image"].cpu().numpy(), g.sdTree.export(), rows


def _worker_drive(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    img, tree, rows = _drive((rank, world, 4))
    np.savez(out % rank, image=img, **{"rec_" + k: v for k, v in rows.items()}, **tree)
    dist.barrier()
    dist.destroy_process_group()


def _worker_drive_batched(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # (a caller-supplied exchange, as bench.py hands the driver libpgsd's own: it is given the accumulators in their 24-byte
    # exchange format and is ORDERED ahead of the image collectives -- exchange_overlap stays off for it)
    from practical_path_guiding_lab_amd.parallel import all_reduce_accumulators
    seen = []

    def exchange(acc):
        seen.append(int(acc.numel()))
        all_reduce_accumulators(acc)
    img, tree, rows = _drive((rank, world, 4), spp_per_pass=1, training_passes_per_launch=5, all_reduce=exchange)
    assert seen and all(n % 1 == 0 for n in seen)
    np.savez(out % rank, image=img, **{"rec_" + k: v for k, v in rows.items()}, **tree)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_driver_with_batched_launches_equals_the_reference_schedule_on_one_rank(tmp_path):
    """main.py's schedule (one-sample training passes) on two ranks, five passes per device launch, the film folded into the
    running mean by pg_film_batched_accumulate on each rank's own rows behind the halo exchange: the image, the tree and
    the logs of ONE rank that launches every pass by itself, bit for bit."""
    world = 2
    out = str(tmp_path / "b%d.npz")
    mp.spawn(_worker_drive_batched, args=(world, 29625, out), nprocs=world, join=True)
    img, tree, rows = _drive(None, spp_per_pass=1)
    for r in range(world):
        got = dict(np.load(out % r))
        np.testing.assert_array_equal(got["image"].view(np.uint32), img.view(np.uint32))
        for key in tree:
            np.testing.assert_array_equal(np.asarray(got[key]).astype(np.float64), np.asarray(tree[key]).astype(np.float64), err_msg=key)
        for key, v in rows.items():
            np.testing.assert_array_equal(got["rec_" + key], v, err_msg=key)


@pytest.mark.gpu
def test_two_rank_driver_in_tile_mode_equals_one_rank(tmp_path):
    """driver.run_guided_render with the film sharded over two ranks (interleaved 4-row bands, gloo,
    both ranks on cuda:0): every rank ends with the image, the SD-tree, and the variance / MSE logs of
    the single-rank run, bit for bit -- lanes gathered before the tent filter, accumulators summed
    before each refine, the stop decision taken on the whole film's sums."""
    world = 2
    out = str(tmp_path / "d%d.npz")
    mp.spawn(_worker_drive, args=(world, 29616, out), nprocs=world, join=True)
    img, tree, rows = _drive(None)
    assert np.isfinite(img).all() and img.max() > 0
    for r in range(world):
        got = dict(np.load(out % r))
        np.testing.assert_array_equal(got["image"].view(np.uint32), img.view(np.uint32))
        for key in tree:
            np.testing.assert_array_equal(np.asarray(got[key]).astype(np.float64), np.asarray(tree[key]).astype(np.float64), err_msg=key)
        for key, v in rows.items():
            np.testing.assert_array_equal(got["rec_" + key], v, err_msg=key)


def _worker_drive_files(rank, world, port, out, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # training stops after iteration 1 (12 spp >= 10): the remaining 48 spp are one final iteration of twelve 4-spp
    # passes that crosses cumm_spp 28 -- an intermediate image (main.py:267-291), i.e. a collective inside the pass loop
    img, tree, rows = _drive((rank, world, 4), out_dir if rank == 0 else None, train_stop_cumm_spp=10)
    np.savez(out % rank, image=img)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_driver_writes_files_on_the_rank_that_was_given_a_directory(tmp_path):
    """out_dir is a per-rank argument.  Rank 0 alone passes one: the run must not hang in the collective
    that assembles an intermediate image (every rank issues it, decided by values all ranks share), rank 0 writes the
    files -- the intermediate blend at 28 spp too -- and both ranks end with the single-rank image."""
    world = 2
    out, out_dir = str(tmp_path / "f%d.npz"), str(tmp_path / "files")
    mp.spawn(_worker_drive_files, args=(world, 29623, out, out_dir), nprocs=world, join=True)
    one_dir = str(tmp_path / "one")
    img, _, _ = _drive(None, one_dir, train_stop_cumm_spp=10)
    for r in range(world):
        np.testing.assert_array_equal(dict(np.load(out % r))["image"].view(np.uint32), img.view(np.uint32))
    names = sorted(os.listdir(out_dir))
    assert names == sorted(os.listdir(one_dir))
    blends = [n for n in names if n.endswith("cumm_spp-28.npy")]
    assert len(blends) == 1
    for n in names:
        if n.endswith(".npy"):
            a, b = np.load(os.path.join(out_dir, n)), np.load(os.path.join(one_dir, n))
            np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32), err_msg=n)


@pytest.mark.gpu
def test_library_exchange_with_one_rank():
    """pg_comm_unique_id / pg_comm_init / pg_allreduce / pg_comm_destroy: RCCL bound at run time, a
    one-rank communicator (what a one-GPU box can form), the all-reduce leaves the accumulators as
    they are, errors are reported before a communicator exists."""
    from practical_path_guiding_lab_amd._native import PgError
    from practical_path_guiding_lab_amd.sdtree import SDTree

    g = SDTree(0)
    g.load(_base_tree())
    g.setIteration(3, False)
    with pytest.raises(PgError):
        g.allReduce()
    rec = synth.records(5000, 11, BB0, BB1)
    g.addDataPropagate({k: torch.from_numpy(v).cuda() for k, v in rec.items()})
    before = g.accumulators().clone()
    kd0, lo0, hi0 = g.exportAccumulators()
    ident = g.commUniqueId()
    assert len(ident) == 128 and any(ident)
    with pytest.raises(PgError):
        g.commInfo()
    g.commInit(1, 0, ident)
    assert g.commInfo() == (1, 0)   # ncclCommCount / ncclCommUserRank read back from RCCL
    g.allReduce()                   # pack (24 B per accumulator) -> ncclAllReduce -> unpack
    torch.cuda.synchronize()
    kd1, lo1, hi1 = g.exportAccumulators()
    # the same values and counts (the limbs come back normalised: the representation may differ, the sums may not)
    np.testing.assert_array_equal(kd0, kd1)
    np.testing.assert_array_equal(lo0, lo1)
    np.testing.assert_array_equal(hi0, hi1)
    assert int(before.abs().sum()) > 0 and int(kd0[0]) == 5000
    a = g.accumulators().clone()
    g.allReduce()                   # normalised limbs are a fixed point of pack -> unpack
    torch.cuda.synchronize()
    assert torch.equal(g.accumulators(), a)
    g.commDestroy()
    with pytest.raises(PgError):
        g.allReduce()


# ---------------------------------------------------------------------------------------------
# bench.py --gpus N from a bare command line: the launcher
# ---------------------------------------------------------------------------------------------
def _bench_module():
    import importlib.util
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
    spec = importlib.util.spec_from_file_location("bench_under_test", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_step_queue_of_eight_ranks_with_steps_that_do_not_fill_the_last_group():
    """bench.py --gpus 8 (tiles): a rank launches the passes of eight steps at once.  The queue that does it, by itself and
    without a GPU: 5 warm-up + 13 timed steps of 16 passes on 8 ranks leave as launches of 5, 8 and 5 steps -- the warm-up
    flushed by itself, the last group smaller -- with seeds that run on without a gap, so that the passes traced are
    exactly those of 18 separate steps; `--shard passes` strides the seeds by the world size."""
    B = _bench_module()
    calls = []
    q = B.StepQueue(lambda n, sd: calls.append((n, sd)), 16, 8, first_seed=252)
    for _ in range(5):
        q.step()
    q.flush()
    for _ in range(13):
        q.step()
    q.flush()
    q.flush()  # (nothing queued: nothing launched)
    assert calls == [(80, 252), (128, 332), (80, 460)] == q.launches
    assert q.seed == 252 + 18 * 16 and sum(n for n, _ in calls) == 18 * 16
    one = B.StepQueue(lambda n, sd: calls.append((n, sd)), 16, 1, first_seed=7, seed_stride=8)   # passes-sharded: group 1
    one.step(); one.step()
    assert one.launches == [(16, 7), (16, 7 + 16 * 8)]


def test_spawn_ranks_gives_every_rank_its_environment_and_relays_rank_zero(tmp_path):
    """bench.spawn_ranks with stand-in rank processes (no GPU): each gets RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR 127.0.0.1 / one common MASTER_PORT; of rank 0's stdout the JSON line is relayed (a banner a library
    printed there -- Gloo does -- goes to stderr), nothing of the other ranks'; exit code 0."""
    import io
    import json
    b = _bench_module()
    child = ("import os, json, sys; r = os.environ['RANK']; "
             "open(sys.argv[1] + r, 'w').write(json.dumps({k: os.environ[k] for k in "
             "('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'HSA_ENABLE_IPC_MODE_LEGACY')})); "
             "print('[Gloo] a banner of rank ' + r); print(json.dumps({'rank': r}))")
    relay = io.StringIO()
    rc = b.spawn_ranks([sys.executable, "-c", child, str(tmp_path / "env")], 3, relay=relay)
    assert rc == 0
    assert relay.getvalue() == '{"rank": "0"}\n'
    envs = [json.load(open(str(tmp_path / "env") + str(r))) for r in range(3)]
    assert [e["RANK"] for e in envs] == ["0", "1", "2"] and [e["LOCAL_RANK"] for e in envs] == ["0", "1", "2"]
    assert all(e["WORLD_SIZE"] == "3" and e["LOCAL_WORLD_SIZE"] == "3" and e["MASTER_ADDR"] == "127.0.0.1" for e in envs)
    assert len({e["MASTER_PORT"] for e in envs}) == 1 and all(e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for e in envs)


def test_spawn_ranks_ends_the_job_when_a_rank_fails():
    """One rank exits with code 7 while the others would run for a minute: the launcher ends them and
    returns 7 within seconds; nothing of rank 0's partial output is mistaken for a result line."""
    import io
    import time
    b = _bench_module()
    child = "import os, sys, time\nif os.environ['RANK'] == '1':\n    sys.exit(7)\ntime.sleep(60)\nprint('never')"
    relay = io.StringIO()
    t0 = time.time()
    rc = b.spawn_ranks([sys.executable, "-c", child], 3, relay=relay, grace_s=5.0)
    assert rc == 7 and time.time() - t0 < 30 and relay.getvalue() == ""


def test_spawn_ranks_ends_its_ranks_when_the_launcher_is_terminated(tmp_path):
    """a harness timeout (SIGTERM to the launcher) must not orphan rank processes that hold GPUs: the launcher
    ends every child by its process id and exits non-zero."""
    import signal
    import subprocess
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pidfile = str(tmp_path / "pid")
    child = f"import os, time; open({pidfile!r} + os.environ['RANK'], 'w').write(str(os.getpid())); time.sleep(120)"
    launcher = ("import sys, importlib.util\n"
                f"spec = importlib.util.spec_from_file_location('b', {os.path.join(root, 'bench.py')!r})\n"
                "b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)\n"
                f"sys.exit(b.spawn_ranks([sys.executable, '-c', {child!r}], 3, grace_s=3.0))\n")
    p = subprocess.Popen([sys.executable, "-c", launcher])
    t0 = time.time()
    while time.time() - t0 < 30 and not all(os.path.exists(pidfile + str(r)) and open(pidfile + str(r)).read() for r in range(3)):
        time.sleep(0.1)
    pids = [int(open(pidfile + str(r)).read()) for r in range(3)]
    p.send_signal(signal.SIGTERM)
    assert p.wait(30) != 0
    time.sleep(0.2)
    for pid in pids:
        alive = True
        try:
            os.kill(pid, 0)
            # (a zombie of another parent cannot be: the launcher waited for its children)
        except ProcessLookupError:
            alive = False
        assert not alive, pid


def test_backend_is_decided_from_the_device_count_alone():
    """auto: nccl when every rank of the node has a device of its own, gloo for a rehearsal on fewer
    devices than ranks -- the same answer on every rank, taken before anything is initialised."""
    import types
    b = _bench_module()
    auto = types.SimpleNamespace(backend="auto")
    assert b.pick_backend(auto, 8, 8) == "nccl" and b.pick_backend(auto, 2, 8) == "nccl"
    assert b.pick_backend(auto, 2, 1) == "gloo" and b.pick_backend(auto, 8, 4) == "gloo"
    assert b.pick_backend(types.SimpleNamespace(backend="gloo"), 2, 8) == "gloo"
    assert b.pick_backend(types.SimpleNamespace(backend="nccl"), 2, 1) == "nccl"


@pytest.mark.gpu
def test_bench_gpus_2_from_a_bare_command_line(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it, on the one-GPU box: the parent starts two
    fresh rank processes before touching the GPU, they rehearse over gloo (two ranks on one device),
    rank 0's JSON line comes back with the honest device count."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    # (three steps: a rank launches the passes of world = 2 steps at once, the odd one leaves by itself when the region is flushed)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--scene", "cornell-box", "--res", "96", "--steps", "3",
           "--warmup", "1", "--train-iters", "3", "--spp-per-pass", "4", "--cpu", "0", "--full-schedule", "0"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    ndev = torch.cuda.device_count()
    assert out["ranks"] == 2 and out["n_gpus"] == min(2, ndev)
    assert out["value"] > 0 and out["scaling"] == "strong"
    if ndev < 2:
        assert "gloo" in out["extra"]["exchange"]
    else:
        assert "RCCL" in out["extra"]["exchange"] or "nccl" in out["extra"]["exchange"]
    c = out["config"]
    assert c["pixels_per_rank_min"] + c["pixels_per_rank_max"] == 96 * 96 and c["pixels_per_rank_min"] > 0
    assert c["steps_per_launch"] == 2 and c["passes_per_launch"] == 8 and out["steps"] == 3
    assert c["paths_per_step"] == 96 * 96 * 4  # (a step is still spp_per_pass passes of the whole film)


def _worker_nccl_one_rank(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["LOCAL_WORLD_SIZE"] = "1"
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    from practical_path_guiding_lab_amd.parallel import (all_reduce_accumulators, all_reduce_sums, init_library_comm,
                                                         max_over_ranks, min_max_over_ranks)
    from practical_path_guiding_lab_amd.sdtree import SDTree

    g = SDTree(0)
    g.load(_base_tree())
    g.setIteration(3, False)
    rec = synth.records(5000, 11, BB0, BB1)
    g.addDataPropagate({k: torch.from_numpy(v).cuda() for k, v in rec.items()})
    before = g.exportAccumulators()
    ok = init_library_comm(g)            # the vote, rank 0's id through broadcast_object_list, ncclCommInitRank, RCCL's own rank count
    if ok:
        ok = g.commInfo() == (1, 0)
        g.allReduce()                    # pg_allreduce: pack -> ncclAllReduce issued by libpgsd.so -> unpack
    all_reduce_accumulators(g.accumulators())   # and the torch.distributed route (a no-op sum with one rank)
    all_reduce_accumulators(g.packAccumulators())
    g.unpackAccumulators()
    torch.cuda.synchronize()
    # (values and counts: the limbs come back normalised from the exchange format)
    same = all(bool((x == y).all()) for x, y in zip(g.exportAccumulators(), before))
    s1, s2 = all_reduce_sums(torch.ones(3, 8, device="cuda"), torch.full((3, 8), 2.0, device="cuda"))
    lo, hi = min_max_over_ranks(7.0)
    t = max_over_ranks(0.25, device="cuda")
    if ok:
        g.commDestroy()
    # a join that FAILS on a rank: the second vote sends every rank to the torch.distributed exchange, nobody raises or hangs
    import warnings
    g2 = SDTree(0)
    g2.load(_base_tree())

    def broken(*a_, **k_):
        raise RuntimeError("ncclCommInitRank: simulated failure")
    g2.commInit = broken
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ok = ok and (init_library_comm(g2) is False)
    # a rank on which RCCL cannot even be LOADED says so in the first vote, before anybody could block in a join
    g3 = SDTree(0)
    g3.load(_base_tree())
    joined = []
    g3.commUniqueId = broken
    g3.commInit = lambda *a_, **k_: joined.append(1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ok = ok and (init_library_comm(g3) is False) and not joined
    dist.barrier()
    dist.destroy_process_group()
    np.save(out, np.array([ok, same, float(s1.sum()) == 24.0 and float(s2.sum()) == 48.0, lo == 7.0 and hi == 7.0, t == 0.25]))


@pytest.mark.gpu
def test_nccl_backend_code_paths_with_one_rank(tmp_path):
    """What a one-GPU box can run of the N > 1 RCCL route: a process group on the `nccl` backend with one rank, the
    collective vote and the id broadcast of parallel.init_library_comm, ncclCommInitRank + pg_allreduce issued by
    libpgsd.so, and every helper bench.py calls with CUDA tensors on that backend (flag devices, reductions of
    timings and pixel counts).  The transport between GPUs stays untested: no such hardware is reachable from here."""
    out = str(tmp_path / "nccl1.npy")
    mp.spawn(_worker_nccl_one_rank, args=(1, 29619, out), nprocs=1, join=True)
    res = np.load(out)
    assert res.all(), res


# ---------------------------------------------------------------------------------------------
# bench.py's in-run counter passes (roofline.traffic): the parsing and the refusals, with a stand-in for rocprofv3
# ---------------------------------------------------------------------------------------------
_FAKE_ROCPROF = r'''#!/usr/bin/env python3
import os, sys
a = sys.argv[1:]
counter, out = a[a.index("--pmc") + 1], a[a.index("-d") + 1]
assert "--" in a and os.path.basename(a[a.index("--") + 1]).startswith("python"), a    # the program itself behind `--`
assert not any(x in a for x in ("--sys-trace", "--kernel-trace", "--stats", "-s", "-r")), a   # counters only, nothing traced
if os.environ.get("FAKE_ROCPROF_FAIL") == counter:
    sys.exit(3)
if os.environ.get("FAKE_ROCPROF_HANG"):
    import subprocess, time
    kid = subprocess.Popen([sys.executable, "-c", "import time; time.sleep(600)"])   # the profiled program, one level down
    open(os.environ["FAKE_ROCPROF_HANG"], "w").write(str(kid.pid))
    time.sleep(600)
os.makedirs(os.path.join(out, "host"), exist_ok=True)
n_guide = int(os.environ.get("FAKE_ROCPROF_GUIDE", "12"))
rows, did = ["Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value"], 0
def put(name, v):
    global did
    did += 1
    rows.append('%d,"%s",%s,%f' % (did, name, counter, v))
base = 100.0 if counter == "FETCH_SIZE" else 10.0
for i in range(8): put("void pg::k_wave_shade<2, false>(pg::RenderArgs)", 7.0)               # training / warm-up: not counted
for i in range(12): put("void pg::k_wave_shade<2, false>(pg::RenderArgs)", base + i)           # 3 steps x 4 launches
for i in range(20): put("void pg::k_wave_shade_a<2, false, false>(pg::RenderArgs)", 1e6)      # the roofline region's other kernels
for i in range(4): put("pg::k_wave_guide(pg::RenderArgs)", 5.0)                                # its warm-up
for i in range(n_guide): put("pg::k_wave_guide(pg::RenderArgs)", 2.0 * base + i)
for i in range(3): put("void pg::k_wave_shade<2, false>(pg::RenderArgs)", 9e9)                 # behind the first k_wave_guide: not `value`'s region
open(os.path.join(out, "host", "1_counter_collection.csv"), "w").write("\n".join(rows) + "\n")
'''


def test_bench_counter_passes_read_their_tables_and_refuse_what_they_cannot_use(tmp_path, monkeypatch):
    """bench.measure_traffic_in_run: two counter passes (FETCH_SIZE, WRITE_SIZE), each a child process of `rocprofv3 --pmc`
    with python3 itself behind `--` and nothing traced; k_wave_guide over the last steps' launches, k_wave_shade over the
    last launches BEFORE the first k_wave_guide; hi = (2 FETCH + WRITE) KiB, lo as counted.  A pass that fails, a table
    with too few launches and a run that is itself profiled give None and the reason (the committed figures then serve)."""
    import argparse
    import time
    B = _bench_module()
    fake = tmp_path / "rocprofv3"
    fake.write_text(_FAKE_ROCPROF)
    fake.chmod(0o755)
    monkeypatch.setenv("PATH", str(tmp_path) + os.pathsep + os.environ["PATH"])
    for k in [k for k in os.environ if k.startswith(("ROCPROF", "ROCP_"))]:
        monkeypatch.delenv(k)
    args = argparse.Namespace(scene="veach-ajar", res=1920, depth=13, spp_per_pass=16, batched=1, train_iters=6, sort=1, in_flight=1)
    want = {"k_wave_guide": 4.0, "k_wave_shade": 4.0}
    fig, note = B.measure_traffic_in_run(args, want, timeout_s=60)
    assert fig is not None, note
    f_sh, w_sh = 100.0 + 5.5, 10.0 + 5.5            # means of base .. base + 11
    f_g, w_g = 200.0 + 5.5, 20.0 + 5.5
    assert fig["k_wave_shade"] == {"hi": int((2 * f_sh + w_sh) * 1024), "lo": int((f_sh + w_sh) * 1024), "atomics": None}
    assert fig["k_wave_guide"] == {"hi": int((2 * f_g + w_g) * 1024), "lo": int((f_g + w_g) * 1024), "atomics": None}
    assert "measured in this run" in note
    # the override reaches traffic_for; without it the committed table answers (or None)
    B.IN_RUN_TRAFFIC.update(fig)
    assert B.traffic_for("k_wave_guide", "no such configuration") == fig["k_wave_guide"]
    B.IN_RUN_TRAFFIC.clear()
    assert B.traffic_for("k_wave_guide", "no such configuration") is None
    monkeypatch.setenv("FAKE_ROCPROF_FAIL", "WRITE_SIZE")
    fig, note = B.measure_traffic_in_run(args, want, timeout_s=60)
    assert fig is None and "WRITE_SIZE exited 3" in note
    monkeypatch.delenv("FAKE_ROCPROF_FAIL")
    monkeypatch.setenv("FAKE_ROCPROF_GUIDE", "7")   # fewer launches than three steps hold
    fig, note = B.measure_traffic_in_run(args, want, timeout_s=60)
    assert fig is None and "11 launches of k_wave_guide" in note   # (4 of its warm-up + 7)
    monkeypatch.delenv("FAKE_ROCPROF_GUIDE")
    # a pass that hangs is ended with everything it started (its process group), and the reason is reported
    pidfile = tmp_path / "kid.pid"
    monkeypatch.setenv("FAKE_ROCPROF_HANG", str(pidfile))
    fig, note = B.measure_traffic_in_run(args, want, timeout_s=3)
    assert fig is None and "did not finish within 3 s" in note
    kid = int(pidfile.read_text())
    for _ in range(50):
        try:
            os.kill(kid, 0)
        except ProcessLookupError:
            break
        st = open(f"/proc/{kid}/stat").read().split()[2] if os.path.exists(f"/proc/{kid}/stat") else "X"
        if st in ("Z", "X"):
            break
        time.sleep(0.1)
    else:
        raise AssertionError("the profiled program of a hung counter pass was left running")
    monkeypatch.delenv("FAKE_ROCPROF_HANG")
    monkeypatch.setenv("ROCPROFILER_SOMETHING", "1")
    fig, note = B.measure_traffic_in_run(args, want, timeout_s=60)
    assert fig is None and "profiler" in note
