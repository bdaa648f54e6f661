"""The guided-render driver: the schedule, stopping policy, image blending and logging of the
reference's main.py (takkasila/practical_path_guiding_lab main.py:92-430) around
PathGuidingIntegrator, without Mitsuba.

Differences that are deliberate and visible in the arguments:
  * `training_spp_per_pass` (main.py:192 hard-codes 1): a pass may trace several samples per pixel
    while training -- the SD-tree result does not depend on how an iteration's samples are split
    into passes (integer accumulation), and bigger wavefronts keep an MI355X busy;
  * the per-pass variance/MSE log of main.py:245-265 is optional (`record_in_iteration`), because
    every entry costs a device->host sync;
  * images are written as PNG (sRGB) + .npy (linear) instead of PNG + EXR.
"""
from __future__ import annotations

import csv
import math
import os
import time
from typing import Callable, Dict, List, Optional

import numpy as np
import torch

from .integrator import PathGuidingIntegrator
from .render import WavefrontScene, render, render_batched


class PerformanceData:
    """common.py:66-97: rows of time, spp, cumm_spp, iteration, variance, mse."""

    FIELDS = ["time", "spp", "cumm_spp", "iteration", "variance", "mse"]

    def __init__(self):
        self.rows: List[List[float]] = []

    def append(self, time=0, spp=0, cumm_spp=0, iteration=0, variance=0, mse=0):
        self.rows.append([time, spp, cumm_spp, iteration, variance, mse])

    def saveToFile(self, fileName: str):
        with open(fileName, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(self.FIELDS)
            w.writerows(self.rows)


def possible_cumm_spp(budget_spp: int) -> List[int]:
    """main.py:105-117."""
    cumm, k, out = 0, 0, []
    while cumm < budget_spp:
        cumm += 2 ** (k + 2)
        out.append(cumm)
        k += 1
    return out


def save_image(path_noext: str, image: torch.Tensor):
    """main.py:277-278, 400-401: the image as `.png` (sRGB) and `.exr` (linear float), plus a `.npy`."""
    from PIL import Image
    from . import exr

    img = image.detach().cpu().numpy().astype(np.float32)
    np.save(path_noext + ".npy", img)
    exr.write_rgb(path_noext + ".exr", img)
    srgb = np.where(img <= 0.0031308, 12.92 * img, 1.055 * np.power(np.clip(img, 0.0031308, None), 1 / 2.4) - 0.055)
    Image.fromarray((np.clip(srgb, 0, 1) * 255 + 0.5).astype(np.uint8)).save(path_noext + ".png")


def load_ground_truth(path: str, width: int, height: int) -> torch.Tensor:
    """main.py:36-41 (`mi.Bitmap(gt).convert(RGB, Float32)`): a ground-truth image as the (3, H*W)
    device tensor computeMSE/computeVariance take. `.exr` (scanline, NONE/ZIP/PIZ) or `.npy`
    (H, W, 3) linear radiance; an image that is an integer multiple of the film size is
    box-downsampled to it (the reference renders at the ground truth's own resolution)."""
    if path.lower().endswith(".exr"):
        from . import exr
        img = exr.read_rgb(path)
    else:
        img = np.load(path)
    img = np.asarray(img, dtype=np.float32)
    if img.ndim != 3 or img.shape[2] != 3:
        raise ValueError(f"ground truth {path}: expected (H, W, 3), got {img.shape}")
    gh, gw = img.shape[:2]
    if (gh, gw) != (height, width):
        if gh % height or gw % width or gh // height != gw // width:
            raise ValueError(f"ground truth {path} is {gw}x{gh}; the film is {width}x{height}")
        f = gh // height
        img = img.reshape(height, f, width, f, 3).astype(np.float64).mean(axis=(1, 3)).astype(np.float32)
    return torch.from_numpy(np.ascontiguousarray(img.reshape(-1, 3).T)).cuda()


def run_guided_render(scene: WavefrontScene, integrator: PathGuidingIntegrator, budget_spp: int,
                      initial_seed: int = 0, ground_truth: Optional[torch.Tensor] = None,
                      sdTreeMaxDepth: int = 20, quadTreeMaxDepth: int = 20, isStoreNEERadiance: bool = True,
                      bsdfSamplingFraction: float = 0.5, batch_spp: int = 4, training_spp_per_pass: int = 1,
                      stable_variance_spp_threshold: int = 256, train_stop_cumm_spp: int = 1000,
                      record_in_iteration: bool = False, out_dir: Optional[str] = None,
                      all_reduce: Optional[Callable[[torch.Tensor], None]] = None,
                      log: Callable[[str], None] = print, shard=None, gt_mask=None,
                      training_passes_per_launch: int = 1, exchange_overlap: Optional[bool] = None) -> Dict:
    """Runs the whole training + rendering schedule; returns the final image, logs and timings.

    training_passes_per_launch = B > 1 (with training_spp_per_pass = 1, the reference's value): B consecutive training
    passes are traced as ONE device pass (pg_pass_params.batched: sample s of the launch is the sample pass seed + s gives
    the pixel) and developed by one film launch -- images, sums, logs and SD-tree are those of B separate one-sample
    passes, bit for bit (tests/test_gpu_render.py); the schedule stays main.py's, only the number of launches changes.

    exchange_overlap: whether a sharded run lets the accumulators' all-reduce travel beside the image sums
    (PathGuidingIntegrator.beginAccumulatorExchange).  None = only when `all_reduce` is the driver's own torch.distributed
    collective, i.e. on the communicator the image sums use too; a caller-supplied `all_reduce` (libpgsd's RCCL
    communicator) is issued at the same point but ORDERED ahead of the image collectives until the two-communicator overlap
    has been validated on a multi-GPU node.

    Multi-GPU (one process per GPU, torch.distributed initialised): shard = (rank, world[, stripe_rows
    [, group]]) makes this rank trace its interleaved bands of the film only (WavefrontScene.set_shard);
    the SD-tree accumulators are summed over the ranks before every refine (exact integers, so every
    rank refines the same tree), every pass's lanes are gathered before the film is developed, and
    variance / MSE / the stop decision are taken on the whole film's sums -- every rank ends with the
    image, the tree and the logs a single rank produces."""
    w, h = scene.film_size
    gather = sums_of = None
    if shard is not None and shard[1] > 1:
        from .parallel import HaloExchange, LaneGather, all_reduce_accumulators, all_reduce_sums
        rank, world = shard[0], shard[1]
        group = shard[3] if len(shard) > 3 else None
        scene.set_shard(rank, world, shard[2] if len(shard) > 2 else 4)
        # bands of rows: every rank develops its own rows and fetches the filter's halo from its neighbours; the
        # images are summed once per iteration.  (One contiguous pixel range per rank: the lanes are gathered whole.)
        gather = HaloExchange(group) if scene.stripe is not None and scene.stripe[0] >= 2 else LaneGather(group)
        sums_of = lambda: all_reduce_sums(integrator.sumL, integrator.sumL2, group)  # noqa: E731
        if all_reduce is None:
            all_reduce = lambda acc: all_reduce_accumulators(acc, group)  # noqa: E731
            if exchange_overlap is None:
                exchange_overlap = True
    whole = gather.reduce_image if hasattr(gather, "reduce_image") else (lambda img: img)
    # Files are written by ONE rank of a sharded run (every rank ends with the same image, tree and logs): the lowest rank
    # that was given an out_dir.  Whether anyone was is agreed on once, here, so that the collectives below never depend on a
    # per-rank argument.
    save_dir, want_blends = out_dir, bool(out_dir)
    if gather is not None:
        from .parallel import lowest_rank_with
        writer = lowest_rank_with(bool(out_dir), group)
        want_blends = writer >= 0
        save_dir = out_dir if writer == rank else None
    bmin, bmax = scene.bbox()
    eps = np.float32(1e-4)  # main.py:55-59
    integrator.setup(numRays=w * h, bbox_min=bmin - eps, bbox_max=bmax + eps, sdTreeMaxDepth=sdTreeMaxDepth,
                     quadTreeMaxDepth=quadTreeMaxDepth, isStoreNEERadiance=isStoreNEERadiance,
                     bsdfSamplingFraction=bsdfSamplingFraction)
    integrator.setGroundTruthMask(gt_mask)  # (the pixels a comparison with the ground truth counts; None: all)
    scene.reserve(integrator, max(batch_spp, training_spp_per_pass, int(training_passes_per_launch)))  # (:93: the record arrays are allocated in setup())
    if save_dir:
        os.makedirs(save_dir, exist_ok=True)
    rec = {k: PerformanceData() for k in ("variance_inIter", "variance_groundTruth_inIter", "mse_groundTruth_inIter",
                                          "variance_endIter", "variance_groundTruth_endIter", "mse_groundTruth_endIter",
                                          "variance_estimated_final")}
    possible = possible_cumm_spp(budget_spp)
    cumm_spp = cumm_spp_prev = image_spp = 0
    remaining = budget_spp
    is_final, is_train, is_clear = False, True, True
    k = 0
    variance_prev = 0.0
    cumm_time = 0.0
    image = prev_iter_image = None
    guided_samples, guided_time = 0, 0.0
    per_iter = []
    while remaining > 0:
        torch.cuda.synchronize()
        t_iter = time.perf_counter()
        if is_clear:
            integrator.resetVarianceCounter()
            image_spp = 0
        if not is_final:
            iter_spp = 2 ** (k + 2)
            if iter_spp == remaining:
                is_final = True
        else:
            iter_spp = remaining
        integrator.setIteration(k, is_final)
        spp_per_pass = batch_spp if is_final else training_spp_per_pass
        n_pass = math.ceil(iter_spp / spp_per_pass)
        done = 0
        curr_iter_image = None
        curr_iter_acc = None
        log(f"Iteration {k}: SPP {iter_spp}, cumm_SPP {cumm_spp}, remaining {budget_spp - cumm_spp}, final {is_final}")
        launch = 1  # one-sample training passes traced per device launch
        if not is_final and spp_per_pass == 1 and not record_in_iteration:
            launch = max(1, int(training_passes_per_launch))
        if launch > 1:
            # `launch` consecutive one-sample passes per device pass; the film folds their images into the iteration's running
            # mean itself (pg_film_batched_accumulate) -- the products and sums the loop below makes of separate images, in
            # the same order, without writing the images
            acc = None
            scale = float(np.float32(1.0 / iter_spp))
            for p in range(0, n_pass, launch):
                nb = min(launch, n_pass - p)
                acc = render_batched(scene, integrator, nb, seed=initial_seed + cumm_spp, gather=gather, mean_of=(acc, scale))
                image_spp += nb
                done += nb
                cumm_spp += nb
            curr_iter_image = acc.reshape(3, h, w).permute(1, 2, 0).contiguous()
            n_pass = 0
        for p in range(n_pass):
            cur = min(spp_per_pass, iter_spp - done)
            img = render(scene, integrator, spp=cur, seed=initial_seed + cumm_spp, gather=gather)  # main.py:218
            wimg = img * float(cur / iter_spp)
            curr_iter_image = wimg if curr_iter_image is None else curr_iter_image + wimg
            if is_final:
                curr_iter_acc = img * cur if curr_iter_acc is None else curr_iter_acc + img * cur
            image_spp += cur
            done += cur
            cumm_spp += cur
            if record_in_iteration:
                el = time.perf_counter() - t_iter + cumm_time
                sums = sums_of() if sums_of else None
                rec["variance_inIter"].append(el, image_spp, cumm_spp, k, variance=integrator.computeVariance(image_spp, sums=sums))
                if ground_truth is not None:
                    rec["variance_groundTruth_inIter"].append(el, image_spp, cumm_spp, k,
                                                              variance=integrator.computeVariance(image_spp, ground_truth, sums))
                    rec["mse_groundTruth_inIter"].append(el, image_spp, cumm_spp, k,
                                                         mse=integrator.computeMSE(image_spp, ground_truth, sums))
            # main.py:267-291, the intermediate image of a long final iteration.  `whole` is a collective when the film is
            # sharded, so whether it is issued may depend only on values every rank holds alike (`want_blends`, not this rank's
            # out_dir: a rank without one would leave the others waiting in the all-reduce)
            if is_final and cumm_spp in possible and prev_iter_image is not None and want_blends:
                cur_cnt = cumm_spp - cumm_spp_prev
                acc_now = whole(curr_iter_acc)
                if save_dir:
                    blend = (acc_now / done * cur_cnt + prev_iter_image * (image_spp - cur_cnt)) / image_spp  # main.py:271-273
                    save_image(os.path.join(save_dir, f"iter-{k}_spp-{image_spp}_cumm_spp-{cumm_spp}"), blend)
        if gather is not None and all_reduce is not None and not is_final:
            # the accumulators start travelling now, beside the image sums and the variance below (is_final is the same
            # on every rank; whether the refine then happens is decided from the whole film's sums, also the same everywhere)
            integrator.beginAccumulatorExchange(all_reduce, overlap=bool(exchange_overlap))
        curr_iter_image = whole(curr_iter_image)  # (sharded with a halo exchange: the ranks' rows become the film, once)
        torch.cuda.synchronize()
        t_render = time.perf_counter() - t_iter
        if k >= 2 or is_final and not is_train:
            guided_samples += w * h * iter_spp
            guided_time += t_render
        if is_final and not is_train and prev_iter_image is not None:
            image = (curr_iter_image * iter_spp + prev_iter_image * (image_spp - iter_spp)) / image_spp  # main.py:287-288
        else:
            image = curr_iter_image
        sums = sums_of() if sums_of else None
        variance = integrator.computeVariance(image_spp, sums=sums)
        variance_gt = integrator.computeVariance(image_spp, ground_truth, sums) if ground_truth is not None else 0.0
        mse_gt = integrator.computeMSE(image_spp, ground_truth, sums) if ground_truth is not None else 0.0
        el = time.perf_counter() - t_iter + cumm_time
        rec["variance_endIter"].append(el, image_spp, cumm_spp, k, variance=variance)
        rec["variance_groundTruth_endIter"].append(el, image_spp, cumm_spp, k, variance=variance_gt)
        rec["mse_groundTruth_endIter"].append(el, image_spp, cumm_spp, k, mse=mse_gt)
        budget_upto_prev = budget_spp - cumm_spp_prev
        variance_current = (variance * image_spp) / budget_upto_prev  # main.py:324-325
        rec["variance_estimated_final"].append(el, image_spp, cumm_spp, k, variance=variance_current)
        log(f"  variance {variance:.6g}  variance_gt {variance_gt:.6g}  mse_gt {mse_gt:.6g}  est.final {variance_current:.6g}  "
            f"render {t_render * 1e3:.1f} ms")
        # main.py:334-377
        next_spp = 2 ** (k + 3)
        remaining = budget_spp - cumm_spp
        stop = (cumm_spp > stable_variance_spp_threshold and variance_current > variance_prev) or cumm_spp >= train_stop_cumm_spp
        was_training = is_train
        if next_spp < remaining:
            if stop:
                is_final, is_train, is_clear = True, False, False
        elif next_spp == remaining:
            is_final = True
            if stop:
                is_train, is_clear = False, False
        else:
            is_final, is_train, is_clear = True, False, False
        t_ref = 0.0
        if is_train and remaining > 0:
            t0 = time.perf_counter()
            integrator.refineAndPrepareSDTreeForNextIteration(all_reduce)  # main.py:383
            torch.cuda.synchronize()
            t_ref = time.perf_counter() - t0
        elif was_training and not is_train:
            log("  -- stop training SDTree --")
        prev_iter_image = image
        torch.cuda.synchronize()
        cumm_time += time.perf_counter() - t_iter
        per_iter.append({"iteration": k, "spp": iter_spp, "render_s": t_render, "refine_s": t_ref})
        if save_dir:
            save_image(os.path.join(save_dir, f"iter-{k}_spp-{image_spp}_cumm_spp-{cumm_spp}"), image)
            integrator.saveSDTreeToFile(os.path.join(save_dir, f"sdtree_iter-{k}.npz"))
            integrator.saveSDTreeOBJ(os.path.join(save_dir, f"kdtree_iter-{k}.obj"))
        variance_prev = variance_current
        k += 1
        cumm_spp_prev = cumm_spp
    if save_dir:
        for name, r in rec.items():
            if r.rows:
                r.saveToFile(os.path.join(save_dir, name + ".csv"))
    return {"image": image, "records": rec, "iterations": per_iter, "cumm_spp": cumm_spp, "time_s": cumm_time,
            "guided_samples": guided_samples, "guided_time_s": guided_time}
