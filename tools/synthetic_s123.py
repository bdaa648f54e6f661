"""SURVEY.md 8(d)'s synthetic kernel-level inputs at their stated sizes, through the C ABI's stand-alone
entry points (no renderer):

  S1 "balanced"  KD complete to depth 12 (4096 leaves) over [0,100]^3, every leaf a complete quadtree of
                 depth 5 (1365 nodes; 5.59 M quadtree nodes), leaf irradiance uniform (0,1]; 2^22 queries
                 (positions uniform in the box, directions uniform on the sphere, PCG32 stream = lane)
  S2 "skewed"    grown on the device by six splat + refine iterations of clustered records with lobed
                 directions (2^19 ... 2^24 records, tests/synth.py), reference thresholds; the same queries
  S3 "splat"     the last S2 record stream (2^24 records) replayed into the S2 topology

For every kernel: time per launch (torch events, 10 launches), units per second, the algorithmic bytes
of SURVEY 8(d) from the measured mean depths (pg_read_depth_counters) and the fraction of the 8 TB/s HBM
peak; beside it the single-threaded CPU oracle on 2^17 of the same units.

    python tools/synthetic_s123.py > gpurun_out/synthetic_s123.json      (GPU box, repo root)
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)

import synth  # noqa: E402
from oracle import pg_oracle as po  # noqa: E402
from practical_path_guiding_lab_amd.sdtree import PCG32Sampler, SDTree  # noqa: E402

BB0, BB1 = [0.0] * 3, [100.0] * 3
NQ = 1 << 22
CPU_N = 1 << 17
PEAK = 8000.0


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in ev) / reps  # ms


def depths(tree, fn):
    tree.enableDepthCounters(True)
    tree.readDepthCounters(reset=True)
    fn()
    torch.cuda.synchronize()
    dc = tree.readDepthCounters(reset=True)
    tree.enableDepthCounters(False)
    return dc


def line(name, n, ms, alg_bytes, cpu_rate=None, extra=None):
    d = {"kernel": name, "units": int(n), "ms": round(ms, 4), "Munits_per_s": round(n / ms / 1e3, 1),
         "alg_bytes_per_launch": int(alg_bytes), "alg_GBps": round(alg_bytes / ms / 1e6, 1),
         "frac_of_hbm_peak": round(alg_bytes / ms / 1e6 / PEAK, 4)}
    # SURVEY 8d's algorithmic bytes price every level of the reference's descents; the jump tables (KD grid, quadtree
    # jump table) turn the top levels into one gather served from L2.  A fraction above 1 is therefore not a bandwidth:
    if d["frac_of_hbm_peak"] > 1.0:
        d["model"] = "served from L2 -- model not applicable (the tables elide the levels the model prices)"
    if cpu_rate is not None:
        d["cpu_oracle_Munits_per_s_1_thread"] = round(cpu_rate, 3)
    if extra:
        d.update(extra)
    return d


def query_suite(tag, tree, otree, p, d):
    """pg_get_leaf_node_index, pg_pdf, pg_sample, pg_guide_bounce on NQ queries of `tree`."""
    n = p.shape[1]
    P, Dr = dev(p), dev(d)
    out = []
    smp = PCG32Sampler(tree, n, seed=0)
    st0 = smp.state.clone()
    # --- leaf index ---
    ms = timed(lambda: tree.getLeafNodeIndex(P))
    dc = depths(tree, lambda: tree.pdf(P, Dr))
    d_kd = dc.kd_levels / max(dc.kd_queries, 1)
    d_q = dc.quad_levels / max(dc.quad_queries, 1)
    t0 = time.perf_counter()
    otree.get_leaf_node_index(p[:, :CPU_N].copy())
    cpu = CPU_N / (time.perf_counter() - t0) / 1e6
    out.append(line(f"{tag} pg_get_leaf_node_index", n, ms, n * 16.0 * d_kd, cpu, {"D_kd": round(d_kd, 3)}))
    # --- pdf ---
    ms = timed(lambda: tree.pdf(P, Dr))
    t0 = time.perf_counter()
    otree.pdf(p[:, :CPU_N].copy(), d[:, :CPU_N].copy())
    cpu = CPU_N / (time.perf_counter() - t0) / 1e6
    out.append(line(f"{tag} pg_pdf", n, ms, n * (16.0 * d_kd + 20.0 * d_q), cpu, {"D_kd": round(d_kd, 3), "D_quad": round(d_q, 3)}))
    # --- sample (+ its pdf) ---
    def do_sample():
        smp.state.copy_(st0)
        tree.sample(P, smp)
    dc = depths(tree, do_sample)
    ds_q = dc.quad_levels / max(dc.quad_queries, 1)
    ms = timed(do_sample)
    st, inc = po.rng_seed(CPU_N, 0)
    t0 = time.perf_counter()
    otree.sample(p[:, :CPU_N].copy(), st, inc)
    cpu = CPU_N / (time.perf_counter() - t0) / 1e6
    out.append(line(f"{tag} pg_sample", n, ms, n * (16.0 * d_kd + 20.0 * ds_q), cpu, {"D_quad": round(ds_q, 3)}))
    # --- the three calls of a bounce: NEE pdf for every lane, half the lanes sample, half evaluate ---
    nee = torch.ones(n, dtype=torch.uint8, device="cuda")
    sel = (torch.arange(n, device="cuda") % 2 + 1).to(torch.uint8)
    dio = Dr.clone()
    run = tree.prepareGuideBounce(P, Dr, nee, sel, dio, smp)
    def do_bounce():
        smp.state.copy_(st0)
        dio.copy_(Dr)
        run()
    dc = depths(tree, do_bounce)
    ms = timed(do_bounce)
    alg = 16.0 * dc.kd_levels + 20.0 * dc.quad_levels
    out.append(line(f"{tag} pg_guide_bounce", n, ms, alg, None,
                    {"D_kd": round(dc.kd_levels / max(dc.kd_queries, 1), 3), "D_quad": round(dc.quad_levels / max(dc.quad_queries, 1), 3),
                     "quad_descents_per_lane": round(dc.quad_queries / n, 3)}))
    return out


def main():
    po.build()
    res = {"note": __doc__.split("\n\n")[0], "hbm_peak_GBps": PEAK, "kernels": []}
    p = synth.positions_uniform(NQ, 3, BB0, BB1)
    d = synth.directions_uniform(NQ, 4)
    # ---- S1 ----
    t0 = time.time()
    o1 = synth.build_balanced(12, 5)
    g1 = SDTree(0)
    g1.load(o1.export())
    s = g1.stats()
    res["S1"] = {"kd_nodes": s.n_kd_nodes, "kd_leaves": s.n_kd_leaves, "quad_nodes": s.n_quad_nodes, "quad_records": s.n_quad_records,
                 "mean_kd_leaf_depth": s.mean_kd_leaf_depth, "mean_quad_leaf_depth": s.mean_quad_leaf_depth, "build_s": round(time.time() - t0, 1)}
    res["kernels"] += query_suite("S1", g1, o1, p, d)
    del g1
    # ---- S2: grown on the device (parity-tested equal to the oracle's refine) ----
    t0 = time.time()
    g2 = SDTree(0)
    g2.setup(BB0, BB1, 0, 0, 20, 20, True, 0.5)
    rec = None
    for k in range(6):
        g2.setIteration(k, False)
        rec = synth.records((1 << 24) >> (5 - k), 77 + 10 * k, BB0, BB1)
        g2.addDataPropagate({kk: dev(v) for kk, v in rec.items()})
        g2.refineAndPrepare()
    s = g2.stats()
    res["S2"] = {"kd_nodes": s.n_kd_nodes, "kd_leaves": s.n_kd_leaves, "quad_nodes": s.n_quad_nodes, "quad_records": s.n_quad_records,
                 "mean_kd_leaf_depth": s.mean_kd_leaf_depth, "mean_quad_leaf_depth": s.mean_quad_leaf_depth, "max_kd_depth": s.max_kd_depth,
                 "max_quad_depth": s.max_quad_depth, "build_s": round(time.time() - t0, 1)}
    o2 = po.OracleTree()
    o2.load(g2.export())
    res["kernels"] += query_suite("S2", g2, o2, p, d)
    # ---- S3: the last record stream replayed into the S2 topology ----
    g2.setIteration(6, False)
    R = {kk: dev(v) for kk, v in rec.items()}
    m = rec["radiance"].shape[0]
    dc = depths(g2, lambda: g2.addDataPropagate(R))
    d_kd = dc.kd_levels / max(dc.kd_queries, 1)
    d_q = dc.quad_levels / max(dc.quad_queries, 1)
    ms = timed(lambda: g2.addDataPropagate(R), reps=5)
    oc = po.OracleTree()
    oc.load(g2.export())
    oc.reset()
    small = {kk: np.ascontiguousarray(v[..., :CPU_N]) for kk, v in rec.items()}
    t0 = time.perf_counter()
    synth.splat(oc, small)
    cpu = CPU_N / (time.perf_counter() - t0) / 1e6
    res["kernels"].append(line("S3 pg_splat", m, ms, m * (16.0 * d_kd + 4.0 + 2.0 * 12.0 * d_q + 48.0), cpu,
                               {"D_kd": round(d_kd, 3), "D_quad": round(d_q, 3), "B_rec": round(16.0 * d_kd + 4.0 + 24.0 * d_q + 48.0, 1)}))
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
