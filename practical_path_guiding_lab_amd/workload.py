"""Synthetic guided-pass workload: the SD-tree side of PathGuidingIntegrator.sample()
(path_guiding_integrator.py:126-431) with the renderer replaced by seeded synthetic surface
points, directions and record contents.

One *pass* = what one `mi.render(spp=1)` call makes the SD-tree do for `num_rays` camera paths:
  per bounce b < max_depth : pg_guide_bounce over the wavefront (NEE pdf + sample-or-pdf, masks
                             model path termination)                       [:244, 301, 307]
  after the loop           : pg_process_and_splat over the dense num_rays*max_depth record buffer
                                                                            [:388-395, 434-500]
Inputs are generated once on the device and stay resident; no ray casting or BSDF work is
included (the renderer substrate is a later scope row).
"""
from __future__ import annotations

import math
from typing import Dict, List

import torch

from .sdtree import PCG32Sampler, SDTree

# cornell-box scene bounds (scenes/cornell-box/scene.xml shapes) +- 1e-4 (main.py:55-59)
CORNELL_BBOX_MIN = (-1.0001, -0.0001, -1.0001)
CORNELL_BBOX_MAX = (1.0001, 2.0001, 1.0001)


def _rand(gen, *shape):
    return torch.rand(shape, generator=gen, device=gen.device, dtype=torch.float32)


def surface_points(gen, n: int, bmin, bmax) -> torch.Tensor:
    """Points on the faces of the scene box and of two inner boxes (cornell-like), planar (3,n)."""
    u = _rand(gen, 5, n)
    lo = torch.tensor(bmin, device=gen.device, dtype=torch.float32)[:, None]
    hi = torch.tensor(bmax, device=gen.device, dtype=torch.float32)[:, None]
    which = (u[0] * 3).to(torch.int64).clamp_(max=2)          # outer box, short box, tall box
    face = (u[1] * 6).to(torch.int64).clamp_(max=5)
    axis, side = face // 2, (face % 2).to(torch.float32)
    c = torch.stack([u[2], u[3], u[4]])                        # (3,n) in [0,1)
    c.scatter_(0, axis[None, :], side[None, :])
    # inner boxes occupy sub-cubes of the unit cube
    blo = torch.tensor([[0.0, 0.0, 0.0], [0.55, 0.0, 0.45], [0.15, 0.0, 0.1]], device=gen.device).T
    bhi = torch.tensor([[1.0, 1.0, 1.0], [0.85, 0.3, 0.8], [0.45, 0.6, 0.45]], device=gen.device).T
    c = blo[:, which] + c * (bhi[:, which] - blo[:, which])
    return (lo + c * (hi - lo)).contiguous()


def unit_dirs(gen, n: int) -> torch.Tensor:
    d = torch.randn((3, n), generator=gen, device=gen.device, dtype=torch.float32)
    return (d / d.norm(dim=0, keepdim=True).clamp_min(1e-20)).contiguous()


def lobe_canonical(gen, n: int) -> torch.Tensor:
    """Canonical (phi, cos) points: 70 % in three tight lobes (light + two bright walls), 30 % uniform."""
    u = _rand(gen, 5, n)
    centres = torch.tensor([[0.25, 0.97], [0.62, 0.40], [0.10, 0.55]], device=gen.device)
    k = (u[0] * 3).to(torch.int64).clamp_(max=2)
    spread = torch.tensor([0.03, 0.08, 0.12], device=gen.device)[k]
    lob = centres[k].T + (u[1:3] - 0.5) * spread
    out = torch.where(u[3] < 0.7, lob, u[1:3])
    return out.remainder(1.0).clamp_(0.0, 1.0).contiguous()


class SyntheticPassWorkload:
    def __init__(self, tree: SDTree, num_rays: int, max_depth: int, seed: int = 1, rank: int = 0,
                 bbox_min=CORNELL_BBOX_MIN, bbox_max=CORNELL_BBOX_MAX, survival: float = 0.8):
        self.tree, self.n, self.depth = tree, int(num_rays), int(max_depth)
        self.bmin, self.bmax = bbox_min, bbox_max
        self.survival = survival
        self.dev = tree.device
        self.gen = torch.Generator(device=self.dev)
        self.gen.manual_seed(seed * 1000003 + rank)
        self.rank = rank
        self.bounce: List[Dict[str, torch.Tensor]] = []
        self.dense: Dict[str, torch.Tensor] = {}
        self.Lfinal = None
        self.sampler = None

    # ---- training: grow a realistic tree with the library itself -----------------------------
    def training_records(self, m: int) -> Dict[str, torch.Tensor]:
        g = self.gen
        u = _rand(g, 3, m)
        radiance = torch.exp2(8.0 * u[0] - 5.0)
        nee = torch.where(u[1] < 0.3, torch.zeros_like(u[1]), torch.exp2(6.0 * u[1] - 3.0))
        return {
            "position": surface_points(g, m, self.bmin, self.bmax),
            "direction": lobe_canonical(g, m),
            "radiance": radiance.contiguous(),
            "woPdf": (0.05 + 0.95 * u[2]).contiguous(),
            "direction_nee": lobe_canonical(g, m),
            "radiance_nee_lum": nee.contiguous(),
        }

    def train(self, iterations: int, records_per_pass: int, all_reduce=None):
        """`iterations` rounds of 2^(k+2) passes (main.py:170) of synthetic records + refine.
        all_reduce: optional callable(int64 tensor) summing the accumulators across ranks."""
        for k in range(iterations):
            self.tree.setIteration(k, False)
            for _ in range(2 ** (k + 2)):
                self.tree.addDataPropagate(self.training_records(records_per_pass))
            if all_reduce is not None:
                all_reduce(self.tree.accumulators())
            self.tree.refineAndPrepare()
        self.tree.setIteration(iterations, False)

    # ---- one resident pass ------------------------------------------------------------------
    def prepare(self):
        g, n, D = self.gen, self.n, self.depth
        alive_u = _rand(g, n)
        self.sampler = PCG32Sampler(self.tree, n, seed=7, lane0=self.rank * n)
        self.bounce = []
        alive_cols = []
        for b in range(D):
            alive = alive_u < (self.survival ** b)
            u = _rand(g, 2, n)
            sel = torch.where(alive, torch.where(u[0] > 0.5, 2, 1), 0).to(torch.uint8)  # :286 next_1d > 0.5 -> tree
            nee = (alive & (u[1] < 0.95)).to(torch.uint8)
            self.bounce.append({
                "p": surface_points(g, n, self.bmin, self.bmax),
                "d_nee": unit_dirs(g, n),
                "d_bsdf": unit_dirs(g, n),
                "dir_io": torch.empty((3, n), dtype=torch.float32, device=self.dev),
                "sel": sel.contiguous(), "nee": nee.contiguous(),
                "pdf_nee": torch.empty(n, dtype=torch.float32, device=self.dev),
                "pdf": torch.empty(n, dtype=torch.float32, device=self.dev),
            })
            alive_cols.append(alive)
        S = n * D
        active = torch.stack(alive_cols, dim=1).reshape(S)  # slot = ray*D + b
        pos = torch.stack([bb["p"] for bb in self.bounce], dim=2).reshape(3, S).contiguous()
        u = _rand(g, 13, S)
        thr_bsdf = (u[0:3] * 0.9 + 0.05)
        self.Lfinal = (_rand(g, 3, n) * 2.0 + 0.5).contiguous()
        thr_rad = (u[3:6] * 0.5).contiguous()
        self.dense = {
            "active": active.to(torch.uint8).contiguous(),
            "position": pos,
            "direction": lobe_canonical(g, S),
            "bsdf": (u[6:9] * 0.8 + 0.2).contiguous(),
            "throughputBsdf": thr_bsdf.contiguous(),
            "throughputRadiance": thr_rad,
            "radiance_nee": torch.where(u[9] < 0.3, torch.zeros_like(u[10:13]), u[10:13]).contiguous(),
            "direction_nee": lobe_canonical(g, S),
            "woPdf": (0.05 + 0.95 * u[9]).contiguous(),
        }
        for k, v in self.dense.items():
            if k != "active":
                v.mul_(active.to(v.dtype))  # dr.zeros leaves unwritten slots at 0 (:116)
        # pre-marshalled launches: per bounce one stream compaction of the live lanes (what a wavefront
        # renderer does between bounces) and one guide_bounce over the compacted list
        self.lane_idx = torch.empty(n, dtype=torch.int32, device=self.dev)
        self.lane_cnt = torch.zeros(2, dtype=torch.int32, device=self.dev)
        self._compact, self._bounce = [], []
        for bb in self.bounce:
            # dir_io holds the BSDF-sampled direction on entry; sel==2 lanes overwrite it with the
            # guided direction and never read it, so the buffer needs no reset between passes
            bb["dir_io"] = bb["d_bsdf"].clone()
            use_compaction = self.compaction
            self._compact.append(self.tree.prepareCompactLanes(bb["sel"], bb["nee"], self.lane_idx, self.lane_cnt)
                                 if use_compaction else None)
            self._bounce.append(self.tree.prepareGuideBounce(
                bb["p"], bb["d_nee"], bb["nee"], bb["sel"], bb["dir_io"], self.sampler, bb["pdf_nee"], bb["pdf"],
                self.lane_idx if use_compaction else None, self.lane_cnt if use_compaction else None))
        self._splat = self.tree.prepareProcessAndSplat(self.n, self.depth, self.Lfinal, self.dense)
        torch.cuda.synchronize()

    compaction = True

    def run_compact(self, b: int):
        if self._compact[b] is not None:
            self._compact[b]()

    def run_bounce(self, b: int):
        self._bounce[b]()

    def run_splat(self):
        self._splat()

    def run_pass(self):
        for b in range(self.depth):
            self.run_compact(b)
            self.run_bounce(b)
        self.run_splat()

    # ---- byte model ---------------------------------------------------------------
    def measure_depths(self) -> Dict[str, float]:
        """Runs one instrumented pass; returns per-launch level counts for the algorithmic-byte model."""
        t = self.tree
        t.enableDepthCounters(True)
        t.readDepthCounters(reset=True)
        out = {"bounce": [], "splat": None}
        for b in range(self.depth):
            self.run_compact(b)
            self.run_bounce(b)
            dc = t.readDepthCounters(reset=True)
            out["bounce"].append((dc.kd_levels, dc.kd_queries, dc.quad_levels, dc.quad_queries))
        self.run_splat()
        dc = t.readDepthCounters(reset=True)
        out["splat"] = (dc.kd_levels, dc.kd_queries, dc.quad_levels, dc.quad_queries)
        t.enableDepthCounters(False)
        return out


def bounce_bytes(kd_levels, kd_queries, quad_levels, quad_queries) -> float:
    """B_bounce summed over a launch: 16 B per KD level + 20 B per quadtree level."""
    return 16.0 * kd_levels + 20.0 * quad_levels


def splat_bytes(kd_levels, records, quad_levels, quad_descents) -> float:
    """B_rec summed over a launch: 16*D_kd + 4 + (4+8)*D_q per descent + 48 B record."""
    return 16.0 * kd_levels + 4.0 * records + 12.0 * quad_levels + 48.0 * records
