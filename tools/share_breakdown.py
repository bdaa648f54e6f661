"""Dev tool (GPU box): where does the time of ONE rank's share go?  Times the kernels of a guided pass over the bands
of rank 0 of `world` (the veach-ajar film, 16 spp per pass), with the live list sorted and not.
    python tools/share_breakdown.py [world]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from practical_path_guiding_lab_amd import scene as S  # noqa: E402
from practical_path_guiding_lab_amd.integrator import PathGuidingIntegrator  # noqa: E402
from practical_path_guiding_lab_amd.render import IndependentSampler, WavefrontScene  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sc = S.veach_ajar(1920, 1080)
for sort in (True, False):
    g = PathGuidingIntegrator({"max_depth": 13, "rr_depth": 8})
    g.setup(1920 * 1080, sc.bbox_min - np.float32(1e-4), sc.bbox_max + np.float32(1e-4), 20, 20, True, 0.5)
    ws = WavefrontScene(sc, sort=sort)
    cumm = 0
    for k in range(5):
        g.setIteration(k, False)
        n = 2 ** (k + 2)
        for i in range(0, n, min(8, n)):
            g.sample(ws, IndependentSampler(min(8, n), cumm + i))
        cumm += n
        g.refineAndPrepareSDTreeForNextIteration()
    g.setIteration(5, False)
    if world > 1:
        ws.set_shard(0, world, 4)
    for i in range(2):
        g.sample(ws, IndependentSampler(16, 900 + i))
    g.sdTree.enableKernelTiming(True)
    g.sdTree.readKernelTiming(reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(5):
        g.sample(ws, IndependentSampler(16, 1000 + i))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5 * 1e3
    kt = g.sdTree.readKernelTiming(reset=True)
    p = max(kt.passes, 1)
    print(f"world {world} sort {sort}: step {dt:.2f} ms; trace {kt.trace_ms / p:.2f} shade {kt.shade_ms / p:.2f} sort {kt.sort_ms / p:.2f} "
          f"splat {kt.splat_ms / p:.2f} finish {kt.finish_ms / p:.2f} tail {kt.tail_ms / p:.2f}; sum "
          f"{(kt.trace_ms + kt.shade_ms + kt.sort_ms + kt.splat_ms + kt.finish_ms + kt.tail_ms) / p:.2f}", flush=True)
    del g, ws
    torch.cuda.empty_cache()
