"""The reference's two other render drivers, kept beside the guided one (driver.py holds main.py's schedule only):

  repeat_high_spp   repeat_high_spp_renderer.py:26-215 -- every iteration's saved SD-tree, frozen, rendered with the same
                    number of samples; the one consumer of loadSDTreeFromFile (path_guiding_integrator.py:597-608)
  run_path_tracing  path_tracing_render.py:24-165 -- the unguided benchmark renderer: what guiding has to beat

Both drive the same PathGuidingIntegrator.sample() through render.render; neither is on the hot path of SURVEY.md 8.
"""
from __future__ import annotations

import os
import time
from typing import Callable, Dict, Optional

import numpy as np
import torch

from .driver import PerformanceData, save_image
from .integrator import PathGuidingIntegrator
from .render import WavefrontScene, render


def repeat_high_spp(scene: WavefrontScene, integrator: PathGuidingIntegrator, tree_dir: str, start_iteration: int,
                    end_iteration: int, max_tree_iteration: int, iter_spp: int, batch_spp: int = 4, initial_seed: int = 0,
                    ground_truth: Optional[torch.Tensor] = None, sdTreeMaxDepth: int = 20, quadTreeMaxDepth: int = 20,
                    isStoreNEERadiance: bool = True, bsdfSamplingFraction: float = 0.5, out_dir: Optional[str] = None,
                    sim_iter: int = 0, gt_mask=None, log: Callable[[str], None] = print) -> Dict:
    """repeat_high_spp_renderer.py:26-215 (doFullSimulation).  Iteration k renders as a final iteration (nothing is
    recorded) with the tree saved after iteration k - 1 (`tree_dir`/sdtree_iter-{k-1}.npz, as run_guided_render leaves
    them; iterations 0 and 1 are unguided, path_guiding_integrator.py:223), passes of `batch_spp`, seeds initial_seed +
    cumulative spp.  Returns and (with out_dir) writes the reference's records -- variance / variance_groundTruth /
    mse_groundTruth `_endIter_high_spp_sim-{sim_iter}.csv` -- and the image of every iteration."""
    w, h = scene.film_size
    bmin, bmax = scene.bbox()
    eps = np.float32(1e-4)
    integrator.setup(numRays=w * h, bbox_min=bmin - eps, bbox_max=bmax + eps, sdTreeMaxDepth=sdTreeMaxDepth,
                     quadTreeMaxDepth=quadTreeMaxDepth, isStoreNEERadiance=isStoreNEERadiance,
                     bsdfSamplingFraction=bsdfSamplingFraction)
    integrator.setGroundTruthMask(gt_mask)
    scene.reserve(integrator, batch_spp)
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
    rec = {k: PerformanceData() for k in ("variance_endIter", "variance_groundTruth_endIter", "mse_groundTruth_endIter")}
    theo_cumm_iter_spp = cumm_spp = 0
    elapsed = 0.0
    images = {}
    for k in range(start_iteration, end_iteration + 1):
        integrator.resetVarianceCounter()
        theo_cumm_iter_spp += 2 ** (k + 1) if k > 0 else 0  # :80-84: 0 4 12 28 ...
        if 0 < k <= max_tree_iteration:  # :87-89
            integrator.loadSDTreeFromFile(os.path.join(tree_dir, f"sdtree_iter-{k - 1}.npz"))
        integrator.setIteration(k, True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        image, done = None, 0
        while done < iter_spp:
            cur = min(batch_spp, iter_spp - done)
            img = render(scene, integrator, spp=cur, seed=initial_seed + cumm_spp)  # :122
            wimg = img * float(cur / iter_spp)
            image = wimg if image is None else image + wimg
            done += cur
            cumm_spp += cur
        torch.cuda.synchronize()
        elapsed += time.perf_counter() - t0
        variance = integrator.computeVariance(done)
        variance_gt = integrator.computeVariance(done, ground_truth) if ground_truth is not None else 0.0
        mse_gt = integrator.computeMSE(done, ground_truth) if ground_truth is not None else 0.0
        rec["variance_endIter"].append(elapsed, done, theo_cumm_iter_spp + done, k, variance=variance)
        rec["variance_groundTruth_endIter"].append(elapsed, done, theo_cumm_iter_spp + done, k, variance=variance_gt)
        rec["mse_groundTruth_endIter"].append(elapsed, done, theo_cumm_iter_spp + done, k, mse=mse_gt)
        log(f"Iteration {k} (frozen tree, {done} spp): variance {variance:.6g}  variance_gt {variance_gt:.6g}  mse_gt {mse_gt:.6g}")
        images[k] = image
        if out_dir:
            save_image(os.path.join(out_dir, f"high_spp_iter-{k}_spp-{done}"), image)
    if out_dir:
        for name, r in rec.items():
            r.saveToFile(os.path.join(out_dir, f"{name}_high_spp_sim-{sim_iter}.csv"))
    return {"records": rec, "images": images, "time_s": elapsed}


def run_path_tracing(scene: WavefrontScene, integrator: PathGuidingIntegrator, target_spp: Optional[int] = None,
                     time_budget_s: Optional[float] = None, chunk_spp: int = 4, initial_seed: int = 0,
                     ground_truth: Optional[torch.Tensor] = None, out_dir: Optional[str] = None, gt_mask=None,
                     log: Callable[[str], None] = print) -> Dict:
    """path_tracing_render.py:24-165: the same sample() without the SD-tree -- here the integrator held at iteration 0 as
    a final iteration (guiding needs iteration > 1, path_guiding_integrator.py:223; nothing is recorded) -- in chunks of
    `chunk_spp` with seeds initial_seed + pass number, until `target_spp` samples or `time_budget_s` seconds (which
    overrides).  Records (time, spp, variance vs ground truth, MSE) after every chunk."""
    if (target_spp is None) == (time_budget_s is None):
        raise ValueError("give target_spp or time_budget_s")
    w, h = scene.film_size
    bmin, bmax = scene.bbox()
    eps = np.float32(1e-4)
    integrator.setup(numRays=w * h, bbox_min=bmin - eps, bbox_max=bmax + eps)
    integrator.setGroundTruthMask(gt_mask)
    integrator.setIteration(0, True)
    integrator.resetVarianceCounter()
    scene.reserve(integrator, chunk_spp)
    rec = PerformanceData()
    image_acc, used, passes = None, 0, 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    elapsed = 0.0
    while (used < target_spp) if time_budget_s is None else (elapsed < time_budget_s):
        cur = chunk_spp if time_budget_s is not None else min(chunk_spp, target_spp - used)
        img = render(scene, integrator, spp=cur, seed=initial_seed + passes)  # :88, 128
        image_acc = img * cur if image_acc is None else image_acc + img * cur
        used += cur
        passes += 1
        mse = integrator.computeMSE(used, ground_truth) if ground_truth is not None else 0.0
        var = integrator.computeVariance(used, ground_truth) if ground_truth is not None else integrator.computeVariance(used)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        rec.append(elapsed, used, used, 0, variance=var, mse=mse)
    image = image_acc / float(used)
    log(f"path tracing: {used} spp in {elapsed:.2f} s = {w * h * used / max(elapsed, 1e-9) / 1e6:.1f} Msamples/s; "
        f"variance {rec.rows[-1][4]:.6g}  mse_gt {rec.rows[-1][5]:.6g}")
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
        save_image(os.path.join(out_dir, f"path_tracing-{used}"), image)
        rec.saveToFile(os.path.join(out_dir, "variance_groundTruth_path_tracing.csv"))
    return {"image": image, "record": rec, "spp": used, "time_s": elapsed}
