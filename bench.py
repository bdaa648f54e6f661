#!/usr/bin/env python
"""bench.py -- guided Msamples/s of the MI355X-native path-guiding integrator.

Workload (BASELINE.json configs[1]): cornell-box 512x512, max_depth 8, guided iterations with the
2^(k+2) spp schedule.  The SD-tree is first trained by really rendering iterations 0..train_iters-1
(untimed); a *step* is then one guided pass of the next iteration: every pixel x spp_per_pass camera
paths through pg_render_pass (camera rays, max_depth bounces with NEE and BSDF/SD-tree one-sample
MIS, record store) followed by record post-processing and the splat into sdTree_current -- i.e.
PathGuidingIntegrator.sample() of the reference, whole.  Everything is resident in HBM.

value     = camera paths per second over all ranks (Msamples/s), wall clock over K steps.
roofline  = dominant kernel of the timed region (HIP events recorded by the library on the launch
            stream): its SD-tree algorithmic bytes per launch / mean launch time vs
            the 8 TB/s HBM peak.
cpu_baseline = the CPU oracle ("port") rendering a guided pass of the same scene and tree at
            1/4 of the pixels on one host core.

`--synthetic` runs the renderer-free hot-path workload instead (seeded synthetic surface points).
Launch:  python bench.py [--gpus 1]
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
    ap.add_argument("--scene", default="cornell-box", choices=["cornell-box", "veach-mis", "torus"],
                    help="cornell-box 512x512 max_depth 8 (BASELINE configs[1], the default), veach-mis 1280x720 "
                         "max_depth 3 (configs[2]) or torus 1024x768 max_depth 30 (configs[4])")
    ap.add_argument("--res", type=int, default=None, help="film width (cornell-box: square film; veach-mis: 16:9; torus: 4:3)")
    ap.add_argument("--depth", type=int, default=None, help="max_depth (default: 8 / 3 / 30)")
    ap.add_argument("--spp-per-pass", type=int, default=8, help="samples per pixel traced by one pass")
    ap.add_argument("--train-iters", type=int, default=6, help="iterations rendered (untimed) to train the SD-tree")
    ap.add_argument("--cpu-res", type=int, default=256, help="film size of the cpu_baseline sample (0 = skip)")
    ap.add_argument("--synthetic", action="store_true", help="renderer-free SD-tree hot-path workload")
    ap.add_argument("--no-compaction", action="store_true", help="(synthetic) mask dead lanes instead of compacting")
    args = ap.parse_args()
    if args.res is None:
        args.res = {"cornell-box": 512, "veach-mis": 1280, "torus": 1024}[args.scene]
    if args.depth is None:
        args.depth = {"cornell-box": 8, "veach-mis": 3, "torus": 30}[args.scene]
    return args


def make_scene(args, width):
    from practical_path_guiding_lab_amd import scene as S

    if args.scene == "veach-mis":
        return S.veach_mis(width, width * 9 // 16, args.depth, 8)
    if args.scene == "torus":
        return S.torus(width, width * 3 // 4, args.depth, 8)
    return S.cornell_box(width, width, args.depth, 8)


def traffic_for(kernel, key):
    """HBM bytes per launch from the committed PMC summary of the same configuration, else None."""
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        j = json.load(open(pmc))
        k = j.get("configs", {}).get(key, {})
        if kernel in k:
            return k[kernel]["hbm_bytes_per_launch"]
    except Exception:
        pass
    return None


def init_dist(args):
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1:
        raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    ndev = torch.cuda.device_count()
    if world > 1 and ndev < world:
        # rehearsal on a box with fewer GPUs than ranks (ranks share devices): RCCL refuses two ranks on
        # one device, so the exchange goes through gloo; the numbers of such a run are not a result
        local_rank = local_rank % ndev
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="gloo")
        return world, rank, local_rank
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    return world, rank, local_rank


def timed_steps(step, steps, warmup, world):
    import torch
    import torch.distributed as dist

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        from practical_path_guiding_lab_amd.parallel import max_over_ranks
        elapsed = max_over_ranks(elapsed, device="cuda")
    return elapsed


# ------------------------------------------------------------------------------------------------
def run_render(args):
    import numpy as np
    import torch
    import torch.distributed as dist
    from practical_path_guiding_lab_amd import scene as S
    from practical_path_guiding_lab_amd import workload as W
    from practical_path_guiding_lab_amd.integrator import PathGuidingIntegrator
    from practical_path_guiding_lab_amd.parallel import all_reduce_accumulators, shard
    from practical_path_guiding_lab_amd.render import IndependentSampler, WavefrontScene

    world, rank, local_rank = init_dist(args)
    # Sharding: the passes of an iteration are independent (main.py:208-218: pass p uses seed
    # initial_seed + cumm_spp), so N ranks trace N consecutive passes of the same film concurrently,
    # each into its own accumulators, and sum them (int64 all-reduce) before the refine.  Weak
    # scaling: every GPU traces the full res x res film per step.
    sc = make_scene(args, args.res)
    integ = PathGuidingIntegrator({"max_depth": args.depth, "rr_depth": 8}, device=local_rank)
    tree = integ.sdTree
    npix = sc.camera.width * sc.camera.height
    integ.setup(npix, sc.bbox_min - np.float32(1e-4), sc.bbox_max + np.float32(1e-4), 20, 20, True, 0.5)  # main.py:56-64
    ws = WavefrontScene(sc)
    my_pixels = npix
    reduce_fn = (lambda acc: all_reduce_accumulators(acc)) if world > 1 else None

    # ---- train: really render iterations 0..train_iters-1 (2^(k+2) spp each, main.py:170) ----
    torch.cuda.synchronize()
    t_train = time.perf_counter()
    cumm = 0
    for k in range(args.train_iters):
        integ.setIteration(k, False)
        iter_spp = 2 ** (k + 2)
        chunk = max(1, min(args.spp_per_pass, iter_spp // world))
        for i in range(iter_spp // chunk):
            if i % world == rank:
                integ.sample(ws, IndependentSampler(chunk, cumm + i * chunk))
        cumm += iter_spp
        integ.refineAndPrepareSDTreeForNextIteration(reduce_fn)
    torch.cuda.synchronize()
    t_train = time.perf_counter() - t_train
    stats = tree.stats()
    k = args.train_iters
    integ.setIteration(k, False)

    seed = [cumm + rank * args.spp_per_pass]

    def step():
        integ.sample(ws, IndependentSampler(args.spp_per_pass, seed[0]))
        seed[0] += args.spp_per_pass * world

    # one instrumented pass for the byte model (depth counters add atomics: not timed)
    tree.enableDepthCounters(True)
    tree.readDepthCounters(reset=True)
    rec_before = int(tree.exportAccumulators()[0][0])  # records counted at the KD root so far
    step()
    dc = tree.readDepthCounters(reset=True)
    tree.enableDepthCounters(False)
    records_per_pass = int(tree.exportAccumulators()[0][0]) - rec_before
    live = tree.renderLiveCounts(args.depth)  # paths alive after each bounce of that pass

    tree.enableKernelTiming(True)
    tree.readKernelTiming(reset=True)
    for _ in range(args.warmup):
        step()
    tree.readKernelTiming(reset=True)
    elapsed = timed_steps(step, args.steps, 0, world)
    kt = tree.readKernelTiming(reset=True)
    tree.enableKernelTiming(False)

    # per-iteration exchange + refine (not part of `value`)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    all_reduce_accumulators(tree.accumulators())
    torch.cuda.synchronize()
    t_allreduce = time.perf_counter() - t1
    t1 = time.perf_counter()
    tree.refineAndPrepare()
    torch.cuda.synchronize()
    t_refine = time.perf_counter() - t1
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return None

    paths_per_step = npix * args.spp_per_pass * world  # all ranks
    value = paths_per_step * args.steps / elapsed / 1e6
    # instrumented pass -> algorithmic bytes per pass on this rank
    # bounce: 16 B per KD level + 20 B per quadtree level; splat: per record 16*D_kd + 4 + 48 + 12 per quadtree level
    # the splat's depths are not separated from the bounce's by the counters, so it is priced with the
    # tree's mean depths over the records actually kept (counted from sdTree_current's leaf counters)
    paths_bytes = 16.0 * dc.kd_levels + 20.0 * dc.quad_levels          # all bounces of one pass
    d_kd = dc.kd_levels / max(dc.kd_queries, 1)
    d_q = dc.quad_levels / max(dc.quad_queries, 1)
    splat_bytes = records_per_pass * (16.0 * d_kd + 4.0 + 48.0 + 12.0 * 2.0 * d_q)  # B_rec with the measured mean depths
    n_b, n_s = max(kt.bounce_launches, 1), max(kt.splat_launches, 1)
    paths_us = 1e3 * kt.bounce_ms / n_b
    splat_us = 1e3 * kt.splat_ms / n_s
    passes = max(kt.passes, 1)
    per_pass = n_b / passes  # bounce launches per pass (= max_depth)
    kern = {
        "k_bounce": {"launches": int(kt.bounce_launches), "avg_us": round(paths_us, 2),
                     "alg_bytes_per_launch": round(paths_bytes / per_pass),
                     "alg_GBps": round(paths_bytes / per_pass / (paths_us * 1e-6) / 1e9, 2) if paths_us else 0.0},
        "k_process_and_splat": {"launches": int(kt.splat_launches), "avg_us": round(splat_us, 2),
                                "records_per_launch": int(records_per_pass), "alg_bytes_per_launch": round(splat_bytes),
                                "alg_GBps": round(splat_bytes / (splat_us * 1e-6) / 1e9, 2) if splat_us else 0.0},
        "k_finish": {"avg_us": round(1e3 * kt.finish_ms / passes, 2)},
    }
    dom = "k_bounce" if kt.bounce_ms >= kt.splat_ms else "k_process_and_splat"
    cfg_key = f"render res={args.res} depth={args.depth} spp={args.spp_per_pass}"
    if args.scene != "cornell-box":
        cfg_key = f"{args.scene} " + cfg_key
    film = f"{sc.camera.width}x{sc.camera.height}"
    what = {"cornell-box": "built-in scene (Mitsuba cornell-box parameters), no textures",
            "veach-mis": "built-in scene (scenes/veach-mis/scene.xml parameters: 3 sphere lamps, 4 Beckmann rough-conductor "
                         "plates, diffuse floor and wall)",
            "torus": "built-in scene (scenes/torus/scene.xml parameters; its five meshes, 23614 triangles, from "
                     "the package data torus_meshes.npz behind a BVH: diffuse donut in a frosted-glass case, aluminium "
                     "brackets, directional light)"}[args.scene]
    roof = {"bound": "hbm", "kernel": dom, "achieved": kern[dom]["alg_GBps"], "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(kern[dom]["alg_GBps"] / HBM_PEAK_GBS, 5),
            "traffic": traffic_for(dom, cfg_key),
            "note": "k_bounce is one whole bounce of the wavefront (ray casting, NEE incl. shadow ray, shading, SD-tree "
                    "queries, record store, state load/store); its algorithmic bytes count only the SD-tree descents "
                    "(16 B/KD level, 20 B/quadtree level, SURVEY 8d). k_process_and_splat is bound by scattered "
                    "atomics, not HBM (DESIGN.md 5)"}
    cpu = cpu_baseline_render(args, tree, sc) if (args.cpu_res > 0 and world == 1) else None
    out = {
        "metric": f"Msamples/s guided, {args.scene} {film} max_depth {args.depth}", "value": round(value, 3), "unit": "Msamples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.scene} {film} per GPU, max_depth {args.depth}, guided iteration "
                               f"{k} (SD-tree trained by rendering iterations 0-{k - 1}), {args.spp_per_pass} spp per pass; "
                               "full PathGuidingIntegrator.sample(): camera rays, NEE, BSDF/SD-tree MIS, record store, "
                               "post-process + splat; " + what,
                   "pixels_per_gpu": my_pixels, "spp_per_pass": args.spp_per_pass,
                   "paths_per_step": paths_per_step, "kd_nodes": stats.n_kd_nodes, "kd_leaves": stats.n_kd_leaves,
                   "quad_records": stats.n_quad_records, "mean_kd_leaf_depth": round(stats.mean_kd_leaf_depth, 3),
                   "mean_quad_leaf_depth": round(stats.mean_quad_leaf_depth, 3),
                   "guided_tree_queries_per_pass": int(dc.quad_queries), "paths_alive_after_bounce": live,
                   "measured_D_kd": round(dc.kd_levels / max(dc.kd_queries, 1), 3),
                   "measured_D_quad": round(dc.quad_levels / max(dc.quad_queries, 1), 3)},
        "roofline": roof, "cpu_baseline": cpu, "kernels": kern,
        "extra": {"train_s": round(t_train, 3), "trained_spp": cumm, "allreduce_ms": round(1e3 * t_allreduce, 3),
                  "refine_ms": round(1e3 * t_refine, 3)},
    }
    if world > 1:
        dist.destroy_process_group()
    return out


def cpu_baseline_render(args, tree, sc_full):
    """The CPU oracle renders one guided pass of the same scene with the same trained tree."""
    from oracle import pg_oracle as po

    po.build()
    pair = po.OracleSDTreePair()
    pair.prev.load(tree.export())
    pair.current.copy_from(pair.prev)
    pair.current.reset()
    sc = make_scene(args, args.cpu_res)
    spp = args.spp_per_pass
    t0 = time.perf_counter()
    po.render_pass(pair, sc, sc.camera, args.depth, 8, args.train_iters, False, 12345, spp, True, 0.5)
    dt = time.perf_counter() - t0
    n = sc.camera.width * sc.camera.height * spp
    return {"value": round(n / dt / 1e6, 4), "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": f"one guided pass of the same scene and SD-tree at {sc.camera.width}x{sc.camera.height} x {spp} spp "
                      f"({n} paths), single-threaded C oracle, {dt:.1f} s"}


# ------------------------------------------------------------------------------------------------
def run_synthetic(args):
    import torch
    import torch.distributed as dist
    from practical_path_guiding_lab_amd.sdtree import SDTree
    from practical_path_guiding_lab_amd import workload as W

    world, rank, local_rank = init_dist(args)
    pixels = args.res * args.res
    rays = pixels * args.spp_per_pass
    tree = SDTree(device=local_rank)
    tree.setup(W.CORNELL_BBOX_MIN, W.CORNELL_BBOX_MAX, rays, args.depth, 20, 20, True, 0.5)
    wl = W.SyntheticPassWorkload(tree, rays, args.depth, seed=1, rank=rank)
    wl.compaction = not args.no_compaction

    def all_reduce(acc):
        if world > 1 and acc.numel():
            dist.all_reduce(acc, op=dist.ReduceOp.SUM)

    wl.train(args.train_iters, records_per_pass=3 * pixels, all_reduce=all_reduce if world > 1 else None)
    torch.cuda.synchronize()
    stats = tree.stats()
    wl.prepare()
    depths = wl.measure_depths()
    D = args.depth
    for _ in range(args.warmup):
        wl.run_pass()
    torch.cuda.synchronize()
    ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(D + 1)]
          for _ in range(args.steps)]
    kk = [0]

    def step():
        k = kk[0]
        for b in range(D):
            wl.run_compact(b)
            ev[k][b][0].record()
            wl.run_bounce(b)
            ev[k][b][1].record()
        ev[k][D][0].record()
        wl.run_splat()
        ev[k][D][1].record()
        kk[0] += 1

    elapsed = timed_steps(step, args.steps, 0, world)
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return None
    value = float(rays) * world * args.steps / elapsed / 1e6
    bounce_ms = sum(ev[k][b][0].elapsed_time(ev[k][b][1]) for k in range(args.steps) for b in range(D))
    splat_ms = sum(ev[k][D][0].elapsed_time(ev[k][D][1]) for k in range(args.steps))
    bounce_bytes = sum(W.bounce_bytes(*x) for x in depths["bounce"])
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
    cfg_key = f"synthetic res={args.res} depth={args.depth} spp={args.spp_per_pass}"
    roof = {"bound": "hbm", "kernel": dom, "achieved": round(kern[dom]["alg_GBps"], 2), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(kern[dom]["alg_GBps"] / HBM_PEAK_GBS, 5), "traffic": traffic_for(dom, cfg_key)}
    kq = sum(x[1] for x in depths["bounce"])
    out = {
        "metric": "Msamples/s guided (SD-tree hot path only, synthetic pass)", "value": round(value, 3),
        "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "C2-synthetic: cornell-box 512x512 pixels x spp_per_pass paths/pass, max_depth 8, SD-tree ops "
                               "only (lane compaction + guide_bounce per bounce, then process_and_splat); no ray casting/BSDF",
                   "pixels_per_gpu": pixels, "spp_per_pass": args.spp_per_pass, "paths_per_pass_per_gpu": rays,
                   "kd_nodes": stats.n_kd_nodes, "quad_records": stats.n_quad_records,
                   "measured_D_kd": round(sum(x[0] for x in depths["bounce"]) / max(kq, 1), 3),
                   "measured_D_quad": round(sum(x[2] for x in depths["bounce"]) / max(sum(x[3] for x in depths["bounce"]), 1), 3),
                   "guided_bounces_per_pass": kq, "records_per_pass": depths["splat"][1]},
        "roofline": roof, "cpu_baseline": None, "kernels": kern,
    }
    if world > 1:
        dist.destroy_process_group()
    return out


def main():
    args = parse()
    out = run_synthetic(args) if args.synthetic else run_render(args)
    if out is not None:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
