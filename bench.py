#!/usr/bin/env python
"""bench.py -- guided-pass throughput of the SD-tree hot path on MI355X.

step      = one guided pass of the hot path over pixels x spp_per_pass camera paths (BASELINE.json
            configs[1] geometry: cornell-box 512x512, max_depth 8; 8 spp per pass): per bounce one
            pg_guide_bounce launch, then one pg_process_and_splat over the dense record buffer.
            Inputs are synthetic (seeded) and resident in HBM before the timed region.
value     = camera paths (samples) completed per second over all ranks, in Msamples/s.
roofline  = algorithmic bytes of the dominant kernel / its mean launch time
            (HIP events on the launch stream inside the timed region) vs the 8 TB/s HBM peak.
cpu_baseline = the CPU oracle (oracle/, "port") running the same pass on one host core.

Launch:  python bench.py [--gpus 1]            or
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pixels", type=int, default=512 * 512, help="pixels per GPU (cornell-box 512x512)")
    ap.add_argument("--spp-per-pass", type=int, default=8,
                    help="samples per pixel traced by one pass (one wavefront launch per bounce)")
    ap.add_argument("--depth", type=int, default=8, help="max_depth (record slots per path)")
    ap.add_argument("--train-iters", type=int, default=6, help="refine iterations used to grow the tree")
    ap.add_argument("--cpu-passes", type=int, default=1, help="oracle passes timed for cpu_baseline (0 = skip)")
    ap.add_argument("--no-events", action="store_true", help="skip per-launch HIP events (pure wall clock)")
    ap.add_argument("--no-compaction", action="store_true", help="mask dead lanes instead of compacting them")
    return ap.parse_args()


def main():
    args = parse()
    # The reference renders 1 spp per pass while training (main.py:192) because each pass keeps a
    # dense numRays*max_depth record buffer in a 2017-size GPU.  With 288 GB per MI355X one pass
    # traces spp_per_pass samples of every pixel: same samples, same tree (integer accumulation is
    # order independent), 8x larger wavefronts.
    args.rays = args.pixels * args.spp_per_pass
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from practical_path_guiding_lab_amd.sdtree import SDTree
    from practical_path_guiding_lab_amd import workload as W

    tree = SDTree(device=local_rank)
    tree.setup(W.CORNELL_BBOX_MIN, W.CORNELL_BBOX_MAX, args.rays, args.depth, 20, 20, True, 0.5)  # main.py:56-64
    wl = W.SyntheticPassWorkload(tree, args.rays, args.depth, seed=1, rank=rank)
    wl.compaction = not args.no_compaction

    def all_reduce(acc):
        if world > 1 and acc.numel():
            dist.all_reduce(acc, op=dist.ReduceOp.SUM)  # RCCL over xGMI: exact int64 sums

    # grow the tree with the library itself: every rank splats its own shard of records, the
    # accumulators are summed, every rank runs the same deterministic refine
    t_train = time.perf_counter()
    wl.train(args.train_iters, records_per_pass=3 * args.pixels, all_reduce=all_reduce if world > 1 else None)
    torch.cuda.synchronize()
    t_train = time.perf_counter() - t_train
    stats = tree.stats()
    wl.prepare()
    depths = wl.measure_depths()

    D = args.depth
    n_launch = D + 1
    use_ev = not args.no_events
    for _ in range(args.warmup):
        wl.run_pass()
    torch.cuda.synchronize()
    ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_launch)]
          for _ in range(args.steps)] if use_ev else None
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        for b in range(D):
            wl.run_compact(b)
            if use_ev:
                ev[k][b][0].record()
            wl.run_bounce(b)
            if use_ev:
                ev[k][b][1].record()
        if use_ev:
            ev[k][D][0].record()
        wl.run_splat()
        if use_ev:
            ev[k][D][1].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-iteration exchange + refine, timed separately (not part of `value`)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    all_reduce(tree.accumulators())
    torch.cuda.synchronize()
    t_allreduce = time.perf_counter() - t1
    t1 = time.perf_counter()
    tree.refineAndPrepare()
    torch.cuda.synchronize()
    t_refine = time.perf_counter() - t1

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    total_samples = float(args.rays) * world * args.steps
    value = total_samples / elapsed / 1e6

    roof = None
    kern = {}
    if use_ev:
        bounce_ms = sum(ev[k][b][0].elapsed_time(ev[k][b][1]) for k in range(args.steps) for b in range(D))
        splat_ms = sum(ev[k][D][0].elapsed_time(ev[k][D][1]) for k in range(args.steps))
        bounce_bytes = sum(W.bounce_bytes(*x) for x in depths["bounce"])          # per pass
        s = depths["splat"]
        splat_bytes = W.splat_bytes(s[0], s[1], s[2], s[3])
        kern = {
            "k_guide_bounce": {"launches": args.steps * D, "avg_us": 1e3 * bounce_ms / (args.steps * D),
                               "alg_bytes_per_launch": bounce_bytes / D,
                               "alg_GBps": bounce_bytes * args.steps / (bounce_ms * 1e-3) / 1e9},
            "k_process_and_splat": {"launches": args.steps, "avg_us": 1e3 * splat_ms / args.steps,
                                    "alg_bytes_per_launch": splat_bytes,
                                    "alg_GBps": splat_bytes * args.steps / (splat_ms * 1e-3) / 1e9},
        }
        dom = "k_guide_bounce" if bounce_ms >= splat_ms else "k_process_and_splat"
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                j = json.load(open(pmc))
                if j.get("rays") == args.rays and j.get("depth") == args.depth and dom in j.get("kernels", {}):
                    traffic = j["kernels"][dom]["hbm_bytes_per_launch"]
            except Exception:
                traffic = None
        roof = {"bound": "hbm", "kernel": dom, "achieved": round(kern[dom]["alg_GBps"], 2), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(kern[dom]["alg_GBps"] / HBM_PEAK_GBS, 5), "traffic": traffic}

    cpu = None
    if args.cpu_passes > 0 and world == 1:
        cpu = cpu_baseline(tree_export=None, wl=wl, args=args)

    kq = sum(x[1] for x in depths["bounce"])
    out = {
        "metric": "Msamples/s guided (SD-tree hot path, synthetic pass)", "value": round(value, 3),
        "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "C2-synthetic: cornell-box 512x512 pixels x spp_per_pass paths/pass, max_depth 8, SD-tree ops only "
                               "(lane compaction + guide_bounce per bounce, then process_and_splat); no ray casting/BSDF",
                   "pixels_per_gpu": args.pixels, "spp_per_pass": args.spp_per_pass, "paths_per_pass_per_gpu": args.rays,
                   "max_depth": D, "train_iters": args.train_iters,
                   "kd_nodes": stats.n_kd_nodes, "kd_leaves": stats.n_kd_leaves, "quad_records": stats.n_quad_records,
                   "mean_kd_leaf_depth": round(stats.mean_kd_leaf_depth, 3),
                   "mean_quad_leaf_depth": round(stats.mean_quad_leaf_depth, 3),
                   "measured_D_kd": round(sum(x[0] for x in depths["bounce"]) / max(kq, 1), 3),
                   "measured_D_quad": round(sum(x[2] for x in depths["bounce"]) / max(sum(x[3] for x in depths["bounce"]), 1), 3),
                   "guided_bounces_per_pass": kq, "records_per_pass": depths["splat"][1]},
        "roofline": roof, "cpu_baseline": cpu, "kernels": kern,
        "extra": {"train_s": round(t_train, 3), "allreduce_ms": round(1e3 * t_allreduce, 3),
                  "refine_ms": round(1e3 * t_refine, 3)},
    }
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(tree_export, wl, args):
    """Times the CPU oracle on the same pass (bounded: --cpu-passes passes of the full batch)."""
    import numpy as np
    from oracle import pg_oracle as po

    po.build()
    o = po.OracleTree()
    o.load(wl.tree.export())
    o.reset()
    n, D = wl.n, wl.depth
    host = []
    for bb in wl.bounce:
        host.append({k: bb[k].cpu().numpy() for k in ("p", "d_nee", "d_bsdf", "sel", "nee")})
    dense = {k: v.cpu().numpy() for k, v in wl.dense.items() if not k.startswith("_")}
    Lfinal = wl.Lfinal.cpu().numpy()
    st, inc = po.rng_seed(n, 7)
    t0 = time.perf_counter()
    for _ in range(args.cpu_passes):
        for b in range(D):
            h = host[b]
            # the reference's three calls (path_guiding_integrator.py:244, 301, 307)
            o.pdf(h["p"], h["d_nee"], h["nee"])
            o.sample(h["p"], st, inc, (h["sel"] == 2).astype(np.uint8))
            o.pdf(h["p"], h["d_bsdf"], (h["sel"] == 1).astype(np.uint8))
        rec = po.process_records(n, D, Lfinal, dense)
        o.add_data_propagate(rec["position"], rec["direction"], rec["radiance"], rec["woPdf"],
                             rec["direction_nee"], rec["radiance_nee_lum"])
    dt = time.perf_counter() - t0
    return {"value": round(n * args.cpu_passes / dt / 1e6, 4), "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": f"{args.cpu_passes} full passes ({n} paths x {D} bounces) of the same synthetic workload, "
                      f"single-threaded C oracle, {dt:.1f} s"}


if __name__ == "__main__":
    main()
