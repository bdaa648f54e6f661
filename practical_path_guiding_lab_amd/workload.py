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


# ================================================================================================
# SURVEY.md 8(d)'s kernel-level synthetic inputs S1 / S2 / S3, generated by the package itself with torch
# (on the GPU in bench.py, where 2^25 records take milliseconds; on the CPU in the tests, which hand the very same
# arrays to the CPU oracle).  Streams are Mitsuba's `independent` sampler -- PCG32 seeded through TEA, the
# published algorithms, in 64-bit integer arithmetic (two's-complement int64 stands in for uint64: products and
# sums wrap alike, right shifts are masked) -- so the inputs are the ones the round-1..3 tools took from the
# oracle's generator (tests/test_host_logic.py compares them bit for bit).
# ================================================================================================
import numpy as np  # noqa: E402

S_BBOX = (0.0, 100.0)          # kdtree.py:700 builds its test tree over [0, 100]^3
S1_KD_DEPTH, S1_QUAD_DEPTH = 12, 5
S_QUERIES = 1 << 22
S2_RECORDS = 1 << 24
S2_ITERATIONS = 6
_M32 = 0xFFFFFFFF


def _tea64(v0: torch.Tensor, v1: torch.Tensor) -> torch.Tensor:
    """Tiny Encryption Algorithm, 4 rounds, on uint32 pairs held in int64 (masked after every step)."""
    v0, v1 = v0.clone(), v1.clone()
    s = 0
    for _ in range(4):
        s = (s + 0x9E3779B9) & _M32
        v0 = (v0 + ((((v1 << 4) & _M32) + 0xA341316C) ^ (v1 + s) ^ ((v1 >> 5) + 0xC8013EA4))) & _M32
        v1 = (v1 + ((((v0 << 4) & _M32) + 0xAD90777D) ^ (v0 + s) ^ ((v0 >> 5) + 0x7E95761E))) & _M32
    return (v0 << 32) | v1   # (bit 63 may be set: the int64 is the uint64's bit pattern)


def _lsr(x: torch.Tensor, k: int) -> torch.Tensor:
    """logical right shift of the uint64 bit pattern held in an int64"""
    return (x >> k) & ((1 << (64 - k)) - 1)


class Pcg32Lanes:
    """n PCG32 streams seeded like Mitsuba's independent sampler: lane i = stream (seed, lane0 + i)."""

    MULT = 0x5851F42D4C957F2D

    def __init__(self, n: int, seed: int, lane0: int = 0, device="cpu"):
        lane = (torch.arange(n, dtype=torch.int64, device=device) + int(lane0)) & _M32
        sd = torch.full((n,), seed & _M32, dtype=torch.int64, device=device)
        initstate, initseq = _tea64(sd, lane), _tea64(lane, sd)
        self.inc = (initseq << 1) | 1
        self.state = torch.zeros(n, dtype=torch.int64, device=device)
        self.next_u32()
        self.state = self.state + initstate
        self.next_u32()

    def next_u32(self) -> torch.Tensor:
        old = self.state
        self.state = old * self.MULT + self.inc
        x = _lsr(_lsr(old, 18) ^ old, 27) & _M32
        rot = _lsr(old, 59)
        return ((x >> rot) | (x << ((32 - rot) & 31))) & _M32

    def next_f32(self) -> torch.Tensor:
        bits = (self.next_u32() >> 9) | 0x3F800000
        return bits.to(torch.int32).view(torch.float32) - 1.0


def s_uniform(n: int, seed: int, draws: int = 1, lane0: int = 0, device="cpu") -> torch.Tensor:
    """(draws, n) fp32 uniforms in [0, 1): lane i draws from stream (seed, lane0 + i)."""
    r = Pcg32Lanes(n, seed, lane0, device)
    return torch.stack([r.next_f32() for _ in range(draws)])


def s_positions_uniform(n: int, seed: int, bbox=S_BBOX, device="cpu") -> torch.Tensor:
    u = s_uniform(n, seed, 3, device=device)
    return (bbox[0] + u * (bbox[1] - bbox[0])).contiguous()


def s_directions_uniform(n: int, seed: int, device="cpu") -> torch.Tensor:
    """Directions uniform on the sphere: canonicalToDir (common.py:100-121) of uniform canonical points, in double (an
    INPUT generator: product and oracle are both handed its result)."""
    u = s_uniform(n, seed, 2, device=device).to(torch.float64)
    cos_t = 2.0 * u[1] - 1.0
    sin_t = torch.sqrt(torch.clamp(1.0 - cos_t * cos_t, min=0.0))
    phi = 2.0 * math.pi * u[0]
    return torch.stack([sin_t * torch.cos(phi), sin_t * torch.sin(phi), cos_t]).to(torch.float32).contiguous()


def s_positions_clustered(n, seed, bbox=S_BBOX, power=3, device="cpu"):
    u = s_uniform(n, seed, 3 * power + 3, device=device)
    t = torch.ones((3, n), dtype=torch.float32, device=device)
    for k in range(power):
        t = t * u[3 * k: 3 * k + 3]
    sgn = torch.where(u[3 * power: 3 * power + 3] < 0.5, -1.0, 1.0).to(torch.float32)
    c = 0.5 + (0.5 * sgn) * t
    return (bbox[0] + c * (bbox[1] - bbox[0])).contiguous()


def s_canonical_lobes(n, seed, shift=0, device="cpu"):
    u = s_uniform(n, seed, 5, device=device)
    centres = np.array([[0.125, 0.75], [0.625, 0.25], [0.375, 0.375], [0.875, 0.875]], np.float32)
    centres = ((centres + np.float32(shift) * np.float32(0.0625)) % np.float32(1.0)).astype(np.float32)
    centres = torch.from_numpy(centres).to(device)
    k = torch.clamp((u[0] * 5.0).to(torch.int64), max=4)
    bg = k == 4
    kk = torch.where(bg, 0, k)
    out = []
    for a in range(2):
        lob = centres[kk, a] + (u[1 + a] * u[3 + a] - 0.25) * 0.0625
        out.append(torch.where(bg, u[1 + a], lob))
    return torch.clamp(torch.stack(out), 0.0, 1.0).contiguous()


def s_records(m: int, seed: int, bbox=S_BBOX, shift: int = 0, device="cpu") -> Dict[str, torch.Tensor]:
    """A record stream shaped like scatterDataIntoSDTree's output (path_guiding_integrator.py:485-497): positions dense
    near the centre of the box, canonical directions in four tight lobes over a uniform background, heavy-tailed
    radiance 2^(8u-4) u', woPdf in [0.05, 1)."""
    u = s_uniform(m, seed + 3, 4, device=device)
    e = torch.floor(u[0] * 8.0).to(torch.int32) - 4
    return {
        "position": s_positions_clustered(m, seed, bbox, device=device), "direction": s_canonical_lobes(m, seed + 1, shift, device),
        "radiance": torch.ldexp(u[1], e), "woPdf": 0.05 + 0.95 * u[2],
        "direction_nee": s_canonical_lobes(m, seed + 2, shift, device),
        "radiance_nee_lum": torch.where(u[3] < 0.25, 0.0, torch.ldexp(u[3], e)).to(torch.float32),
    }


def s2_record_stream(k: int, device="cpu") -> Dict[str, torch.Tensor]:
    """Records of iteration k of S2's six splat + refine iterations: 2^19 ... 2^24 of them."""
    return s_records(S2_RECORDS >> (S2_ITERATIONS - 1 - k), 77 + 10 * k, device=device)


def s1_balanced_tree(kd_depth: int = S1_KD_DEPTH, quad_depth: int = S1_QUAD_DEPTH, seed: int = 1234, bbox=S_BBOX) -> dict:
    """S1 in the reference's 23-key schema (kdtree.py:575-602), built directly: a KD tree complete to `kd_depth` over
    bbox^3 in the reference's node numbering (round r splits the 2^r leaves of depth r in ascending order, children
    appended at old + 2 i, old + 2 i + 1; the left child keeps its parent's quadtree, the right one gets tree 2^r + i:
    kdtree.py:243-245, 316-323), every leaf owning a complete quadtree of `quad_depth` in the canonical arena (roots,
    then level by level, four consecutive children per node in quadrant order 1..4, quadtree.py:110-175, 695-851);
    leaf irradiance 1 - u (PCG32 stream = node index, seed 1234), inner nodes c1 + c2 + c3 + c4 in fp32, that order."""
    lo, hi = np.float32(bbox[0]), np.float32(bbox[1])
    n_kd = (1 << (kd_depth + 1)) - 1
    bmin = np.zeros((n_kd, 3), np.float32)
    bmax = np.zeros((n_kd, 3), np.float32)
    depth = np.zeros(n_kd, np.uint32)
    left = np.zeros(n_kd, np.uint32)
    right = np.zeros(n_kd, np.uint32)
    qroot = np.zeros(n_kd, np.uint32)
    bmin[0], bmax[0] = lo, hi
    for d in range(kd_depth):
        first, cnt = (1 << d) - 1, 1 << d
        par = np.arange(first, first + cnt)
        l = (first + cnt) + 2 * np.arange(cnt)
        r = l + 1
        axis = d % 3
        mid = ((bmin[par, axis] + bmax[par, axis]) / np.float32(2.0)).astype(np.float32)
        for ch in (l, r):
            bmin[ch], bmax[ch], depth[ch] = bmin[par], bmax[par], d + 1
        bmax[l, axis] = mid
        bmin[r, axis] = mid
        left[par], right[par] = l, r
        qroot[l] = qroot[par]
        qroot[r] = cnt + np.arange(cnt)
    is_leaf = depth == kd_depth
    R = 1 << kd_depth
    per_tree = sum(4 ** l for l in range(quad_depth + 1))
    n_q = R * per_tree
    base = [R * sum(4 ** j for j in range(l)) for l in range(quad_depth + 2)]  # first node of level l
    q_depth = np.zeros(n_q, np.uint32)
    q_min = np.zeros((n_q, 2), np.float32)
    q_max = np.ones((n_q, 2), np.float32)
    q_child = np.zeros((4, n_q), np.uint32)
    for l in range(quad_depth):
        n_l = R * 4 ** l
        par = base[l] + np.arange(n_l)
        mid = ((q_min[par] + q_max[par]) / np.float32(2.0)).astype(np.float32)
        for c in range(4):
            ch = base[l + 1] + 4 * np.arange(n_l) + c
            q_child[c, par] = ch
            q_depth[ch] = l + 1
            # quadtree.py:153-175: child 1 = [mid, max], 2 = x [min, mid] y [mid, max], 3 = [min, mid], 4 = x [mid, max] y [min, mid]
            x_hi, y_hi = c in (0, 3), c in (0, 1)
            q_min[ch, 0] = mid[:, 0] if x_hi else q_min[par, 0]
            q_max[ch, 0] = q_max[par, 0] if x_hi else mid[:, 0]
            q_min[ch, 1] = mid[:, 1] if y_hi else q_min[par, 1]
            q_max[ch, 1] = q_max[par, 1] if y_hi else mid[:, 1]
    q_leaf = q_depth == quad_depth
    irr = np.zeros(n_q, np.float32)
    u = s_uniform(n_q, seed, 1)[0].numpy()
    irr[q_leaf] = (np.float32(1) - u[q_leaf]).astype(np.float32)
    for l in range(quad_depth - 1, -1, -1):
        par = base[l] + np.arange(R * 4 ** l)
        s = irr[q_child[0, par]]
        for c in (1, 2, 3):
            s = (s + irr[q_child[c, par]]).astype(np.float32)
        irr[par] = s
    return {
        "kdtree_maxLeafSize": np.float64(1.0), "kdtree_maxDepth": np.int64(max(kd_depth, 1)),
        "kdtree_bbox_min": bmin, "kdtree_bbox_max": bmax, "kdtree_depth": depth,
        "kdtree_vertCount": np.zeros(n_kd, np.float32), "kdtree_isLeaf": is_leaf,
        "kdtree_quadTreeRootIndex": qroot, "kdtree_child_left_index": left, "kdtree_child_right_index": right,
        "quadtree_maxDepth": np.int64(max(quad_depth, 1)), "quadtree_isStoreNEERadiance": np.bool_(True),
        "quadtree_rootNodeIndex": np.arange(R, dtype=np.uint32), "quadtree_bbox_min": q_min, "quadtree_bbox_max": q_max,
        "quadtree_depth": q_depth, "quadtree_irradiance": irr, "quadtree_isLeaf": q_leaf,
        "quadtree_refinementThreshold": np.full(n_q, np.inf, np.float32),  # (the initial root's, copied by every split: quadtree.py:355-359, 133)
        "quadtree_child_1_index": q_child[0], "quadtree_child_2_index": q_child[1],
        "quadtree_child_3_index": q_child[2], "quadtree_child_4_index": q_child[3],
    }
