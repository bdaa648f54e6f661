"""Dev tool (GPU box, repo root): how evenly do the interleaved bands of `world` ranks load a GPU?  Trains the
veach-ajar tree once on the full film, then times one guided pass of every rank's share on this one GPU;
the N-GPU step takes as long as the slowest share.  A step is `spp` batched one-sample passes (bench.py's).  Also printed: a
whole training iteration (128 spp = 8 such steps) with the per-rank FIXED costs that do not shrink with N -- the refine
(measured here; every rank refines the same tree) and the accumulators' all-reduce (an ESTIMATE: bytes x 2 (N-1)/N over one
153 GB/s xGMI link per direction of a ring -- no multi-GPU hardware is reachable from here).
    python tools/stripe_balance.py [world] [rows] [spp] [passes in flight: 1 | 2] [passes per launch of a rank's share]
The last argument (default: spp) is what bench.py --gpus N does since round 4: a rank launches spp x N of the iteration's
one-sample passes at once, so that its launch is as big as the one-GPU launch of the whole film."""
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
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 4
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 16
sc = S.veach_ajar(1920, 1080)
g = PathGuidingIntegrator({"max_depth": 13, "rr_depth": 8})
g.setup(1920 * 1080, sc.bbox_min - np.float32(1e-4), sc.bbox_max + np.float32(1e-4), 20, 20, True, 0.5)
in_flight = int(sys.argv[4]) if len(sys.argv) > 4 else 1
share_spp = int(sys.argv[5]) if len(sys.argv) > 5 else spp
ws = WavefrontScene(sc, in_flight=in_flight)
cumm = 0
for k in range(5):
    g.setIteration(k, False)
    n = 2 ** (k + 2)
    for i in range(0, n, min(8, n)):
        g.sample(ws, IndependentSampler(min(8, n), cumm + i, batched=True))
    cumm += n
    g.refineAndPrepareSDTreeForNextIteration()
g.setIteration(5, False)


def timed(n=spp):
    """ms per `spp` passes, launched n at a time"""
    g.sample(ws, IndependentSampler(n, 999, batched=True))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(6):
        g.sample(ws, IndependentSampler(n, 1000 + i * n, batched=True))
    ws.join()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 6 * 1e3 * spp / n


full = timed()
shares = []
for r in range(world):
    ws.set_shard(r, world, rows)
    shares.append(timed(share_spp))
ws.set_shard(0, 1)
print(f"full film {full:.2f} ms; shares of {world} ranks ({rows}-row bands, {spp} spp" + (f", launched {share_spp} passes at a time" if share_spp != spp else "") + "): " + " ".join(f"{t:.2f}" for t in shares))
print(f"slowest share {max(shares):.2f} ms -> speed-up {full / max(shares):.2f}x of {world} (mean share {np.mean(shares):.2f} ms: "
      f"{full / np.mean(shares):.2f}x without imbalance)")

# a whole training iteration of 128 spp with the fixed costs of every rank
torch.cuda.synchronize()
t0 = time.perf_counter()
g.refineAndPrepareSDTreeForNextIteration()
torch.cuda.synchronize()
refine_ms = (time.perf_counter() - t0) * 1e3
acc_bytes = int(g.sdTree.packAccumulators().numel()) * 8  # (what travels since round 5: the 24-byte exchange format)
steps = 128 // spp
allreduce_ms = acc_bytes * 2.0 * (world - 1) / world / 153e9 * 1e3
t1 = steps * full + refine_ms
tn = steps * max(shares) + allreduce_ms + refine_ms
print(f"iteration of 128 spp: one rank {t1:.1f} ms ({steps} steps + refine {refine_ms:.2f} ms); {world} ranks {tn:.1f} ms "
      f"({steps} x {max(shares):.2f} + all-reduce of {acc_bytes / 1e6:.0f} MB ~ {allreduce_ms:.2f} ms (estimate) + refine {refine_ms:.2f} ms) "
      f"-> {t1 / tn:.2f}x of {world}; with the all-reduce issued beside the image sums (beginAccumulatorExchange) at best "
      f"{t1 / (steps * max(shares) + refine_ms):.2f}x")
