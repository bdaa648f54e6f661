"""Seeded synthetic inputs for the SD-tree path (S1 balanced, S2 skewed, S3 splat).

Test infrastructure.  All randomness comes from the oracle's PCG32 (oracle/pgo_math.h), so the
streams do not depend on the numpy version; only exactly-rounded fp32 operations are applied
on top (scale/shift), so every array here is bit-reproducible.
"""
from __future__ import annotations

import numpy as np

from oracle import pg_oracle as po


def uniform(n: int, seed: int, draws: int = 1, lane0: int = 0) -> np.ndarray:
    """(draws, n) fp32 uniforms in [0,1): lane i uses PCG32 stream (seed, lane0+i)."""
    st, inc = po.rng_seed(n, seed, lane0)
    out = np.empty((draws, n), np.float32)
    for k in range(draws):
        out[k] = po.rng_next_f32(st, inc)
    return out


def positions_uniform(n, seed, bbox_min, bbox_max):
    u = uniform(n, seed, 3)
    lo = np.asarray(bbox_min, np.float32)[:, None]
    hi = np.asarray(bbox_max, np.float32)[:, None]
    return (lo + u * (hi - lo)).astype(np.float32)


def positions_clustered(n, seed, bbox_min, bbox_max, power=3):
    """Skewed positions: product of `power` uniforms per axis pulls mass towards bbox_min corner,
    mirrored per octant by a sign draw, centred -> dense near the centre (exact fp32 ops only)."""
    u = uniform(n, seed, 3 * power + 3)
    lo = np.asarray(bbox_min, np.float32)[:, None]
    hi = np.asarray(bbox_max, np.float32)[:, None]
    t = np.ones((3, n), np.float32)
    for k in range(power):
        t = (t * u[3 * k : 3 * k + 3]).astype(np.float32)
    sgn = np.where(u[3 * power : 3 * power + 3] < 0.5, np.float32(-1), np.float32(1))
    c = (np.float32(0.5) + np.float32(0.5) * sgn * t).astype(np.float32)  # in [0,1], dense near 0.5
    return (lo + c * (hi - lo)).astype(np.float32)


def canonical_uniform(n, seed):
    return uniform(n, seed, 2)


def canonical_lobes(n, seed, shift=0):
    """Mixture of 4 tight lobes + uniform background in canonical [0,1)^2.  `shift` moves the
    lobes (in sixteenths) so that successive iterations make earlier refinement obsolete (merges)."""
    u = uniform(n, seed, 5)
    centres = np.array([[0.125, 0.75], [0.625, 0.25], [0.375, 0.375], [0.875, 0.875]], np.float32)
    centres = ((centres + np.float32(shift) * np.float32(0.0625)) % np.float32(1.0)).astype(np.float32)
    k = np.minimum((u[0] * np.float32(5)).astype(np.int32), 4)
    out = np.empty((2, n), np.float32)
    bg = k == 4
    kk = np.where(bg, 0, k)
    spread = np.float32(0.0625)
    for a in range(2):
        lob = centres[kk, a] + (u[1 + a] * u[3 + a] - np.float32(0.25)) * spread
        out[a] = np.where(bg, u[1 + a], lob).astype(np.float32)
    return np.clip(out, np.float32(0), np.float32(1)).astype(np.float32)


def directions_uniform(n, seed):
    """Unit-ish directions via the oracle's canonicalToDir of uniform canonical points."""
    return po.canonical_to_dir(canonical_uniform(n, seed))


def records(m, seed, bbox_min, bbox_max, skew=True, shift=0):
    """A record stream shaped like scatterDataIntoSDTree's output (path_guiding_integrator.py:485-497)."""
    pos = positions_clustered(m, seed, bbox_min, bbox_max) if skew else positions_uniform(m, seed, bbox_min, bbox_max)
    d = canonical_lobes(m, seed + 1, shift) if skew else canonical_uniform(m, seed + 1)
    dn = canonical_lobes(m, seed + 2, shift) if skew else canonical_uniform(m, seed + 2)
    u = uniform(m, seed + 3, 4)
    # heavy-tailed positive radiance: 2^(8u-4) * u'  (exact: ldexp + multiply)
    e = np.floor(u[0] * np.float32(8)).astype(np.int32) - 4
    radiance = np.ldexp(u[1], e).astype(np.float32)
    radiance_nee = np.where(u[3] < 0.25, np.float32(0), np.ldexp(u[3], e)).astype(np.float32)
    wo_pdf = (np.float32(0.05) + np.float32(0.95) * u[2]).astype(np.float32)
    return {
        "position": pos, "direction": d, "radiance": radiance, "woPdf": wo_pdf,
        "direction_nee": dn, "radiance_nee_lum": radiance_nee,
    }


def splat(tree, rec):
    tree.add_data_propagate(rec["position"], rec["direction"], rec["radiance"], rec["woPdf"],
                            rec["direction_nee"], rec["radiance_nee_lum"])


def build_balanced(kd_depth: int, quad_depth: int, seed: int = 1234, bbox=(0.0, 100.0)):
    """S1: complete KD tree to `kd_depth`, every leaf owning a complete quadtree of `quad_depth`;
    leaf irradiance uniform (0,1], inner = c1+c2+c3+c4 (fp32, that order), bottom-up."""
    t = po.OracleTree()
    t.setup([bbox[0]] * 3, [bbox[1]] * 3, max(kd_depth, 1), max(quad_depth, 1), True)
    for _ in range(quad_depth):
        t.quad_split(t.quad_all_leaves())
    for _ in range(kd_depth):
        t.kd_split(t.kd_all_leaves())
    t.clean_unused_quadtree()
    d = t.export()
    n = d["quadtree_depth"].shape[0]
    irr = np.zeros(n, np.float32)
    leaf = d["quadtree_isLeaf"]
    u = uniform(n, seed, 1)[0]
    irr[leaf] = (np.float32(1) - u[leaf]).astype(np.float32)  # (0,1]
    depth = d["quadtree_depth"]
    cs = [d["quadtree_child_%d_index" % i] for i in (1, 2, 3, 4)]
    for lv in range(quad_depth - 1, -1, -1):
        sel = np.nonzero((depth == lv) & ~leaf)[0]
        s = irr[cs[0][sel]]
        for j in (1, 2, 3):
            s = (s + irr[cs[j][sel]]).astype(np.float32)
        irr[sel] = s
    d["quadtree_irradiance"] = irr
    t.load(d)
    return t


def build_skewed(m: int, iterations: int, seed: int = 77, bbox=(0.0, 100.0), kd_max_depth=20,
                 quad_max_depth=20, c_scale=None):
    """S2: grow an SD-tree pair by `iterations` splat+refine rounds of m*2^k records each."""
    pair = po.OracleSDTreePair()
    pair.setup([bbox[0]] * 3, [bbox[1]] * 3, kd_max_depth, quad_max_depth, True)
    for k in range(iterations):
        rec = records(m << k, seed + 10 * k, [bbox[0]] * 3, [bbox[1]] * 3)
        splat(pair.current, rec)
        pair.refine_and_prepare(k)
    return pair
