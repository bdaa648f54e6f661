"""Dev tool (GPU box, repo root): the statistics behind the bounds of test_veach_ajar_agrees_with_the_tungsten_ground_truth --
the same 1020-spp schedule at 320x180 with several seeds: per seed the mean, mean |r - 1| and max |r - 1| of the 15x20 block
ratios image / ground truth, and per block the spread over the seeds (the estimator's own noise) next to its mean offset
(what the missing teapots and the 720p ground truth filtered to the film's size leave).
    python tools/ajar_block_spread.py [n_seeds]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)

import test_gpu_render as T  # noqa: E402
from practical_path_guiding_lab_amd import scene as S  # noqa: E402
from practical_path_guiding_lab_amd.driver import load_ground_truth, run_guided_render  # noqa: E402
from practical_path_guiding_lab_amd.integrator import PathGuidingIntegrator  # noqa: E402
from practical_path_guiding_lab_amd.render import WavefrontScene  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
gtf = os.path.join(ROOT, "tests", "golden", "veach_ajar_gt_320x180_f16.npy")
gtn = np.load(gtf).astype(np.float64)
mask = S.veach_ajar_mask(320, 180)
rows = []
for seed in range(3, 3 + n):
    sc = S.veach_ajar(320, 180)
    gt = load_ground_truth(gtf, 320, 180)
    g = PathGuidingIntegrator({"max_depth": 13, "rr_depth": 8})
    res = run_guided_render(WavefrontScene(sc), g, 1020, initial_seed=seed, ground_truth=gt, training_spp_per_pass=4, log=lambda s: None)
    img = np.where(mask[..., None], res["image"].cpu().numpy().astype(np.float64), gtn)
    r = T._block_ratios(img, gtn, 15, 20)
    rows.append(r)
    print(f"seed {seed}: blocks {r.size}  mean {r.mean():.4f}  mean|r-1| {np.abs(r - 1).mean():.4f}  max|r-1| {np.abs(r - 1).max():.4f}  "
          f"image mean {img[mask].mean() / gtn[mask].mean() - 1:+.4f}", flush=True)
R = np.array(rows)
off, sd = R.mean(axis=0) - 1, R.std(axis=0, ddof=1)
k = np.argsort(-np.abs(R - 1).max(axis=0))[:8]
print("worst blocks: mean offset / std over seeds / worst seed")
for i in k:
    print(f"  block {i}: offset {off[i]:+.4f}  std {sd[i]:.4f}  worst {np.abs(R[:, i] - 1).max():.4f}")
print(f"over all blocks: |offset| mean {np.abs(off).mean():.4f} max {np.abs(off).max():.4f}; std mean {sd.mean():.4f} max {sd.max():.4f}; "
      f"max over seeds of max|r-1|: {np.abs(R - 1).max():.4f}; of mean|r-1|: {np.abs(R - 1).mean(axis=1).max():.4f}; of |mean-1|: {np.abs(R.mean(axis=1) - 1).max():.4f}")
