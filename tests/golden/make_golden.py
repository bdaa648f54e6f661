"""Generates tests/golden/sdtree_golden.npz from the CPU oracle (the reference itself holds no
vectors and cannot run here: DESIGN.md section 6).  Inputs are the seeded streams of tests/synth.py.

    python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import synth  # noqa: E402
from oracle import pg_oracle as po  # noqa: E402

BB0, BB1 = [0.0] * 3, [100.0] * 3
KEYS = ("kdtree_bbox_min", "kdtree_bbox_max", "kdtree_depth", "kdtree_vertCount", "kdtree_isLeaf",
        "kdtree_quadTreeRootIndex", "kdtree_child_left_index", "kdtree_child_right_index",
        "quadtree_rootNodeIndex", "quadtree_bbox_min", "quadtree_bbox_max", "quadtree_depth",
        "quadtree_irradiance", "quadtree_isLeaf", "quadtree_refinementThreshold", "quadtree_child_1_index",
        "quadtree_child_2_index", "quadtree_child_3_index", "quadtree_child_4_index")


def tree_digest(export: dict) -> str:
    h = hashlib.sha256()
    for k in KEYS:
        a = np.ascontiguousarray(export[k])
        if a.dtype == bool:
            a = a.astype(np.uint8)
        h.update(k.encode())
        h.update(str(a.shape).encode())
        h.update(a.tobytes())
    return h.hexdigest()


def lifecycle(splat_fn, refine_fn, export_fn, iterations=4, m0=1 << 15, seed=2024):
    """Runs the golden lifecycle through callables so the same recipe drives oracle and device."""
    digests = []
    for k in range(iterations):
        rec = synth.records(m0 << k, seed + 10 * k, BB0, BB1, shift=k)
        splat_fn(rec)
        refine_fn(k)
        digests.append(tree_digest(export_fn()))
    return digests


def queries(n=256):
    return synth.positions_uniform(n, 99, BB0, BB1), synth.directions_uniform(n, 98)


def main():
    pair = po.OracleSDTreePair()
    pair.setup(BB0, BB1, 20, 20, True)
    digests = lifecycle(lambda r: synth.splat(pair.current, r), lambda k: pair.refine_and_prepare(k),
                        lambda: pair.prev.export())
    e = pair.prev.export()
    p, d = queries()
    n = p.shape[1]
    st, inc = po.rng_seed(n, 17)
    sd, spdf = pair.prev.sample(p, st, inc)
    pdf = pair.prev.pdf(p, d)
    leaf = pair.prev.get_leaf_node_index(p)
    rec = synth.records(4096, 31337, BB0, BB1)
    synth.splat(pair.current, rec)
    out = {
        "digests": np.array(digests),
        "n_kd": np.int64(e["kdtree_depth"].shape[0]), "n_quad": np.int64(e["quadtree_depth"].shape[0]),
        "n_roots": np.int64(e["quadtree_rootNodeIndex"].shape[0]),
        "sample_dir": sd, "sample_pdf": spdf, "pdf": pdf, "leaf": leaf, "rng_state_after": st,
        "splat_kd_count": pair.current.kd_column("count"),
        "splat_acc_lo": pair.current.quad_column("acc_lo"), "splat_acc_hi": pair.current.quad_column("acc_hi"),
    }
    np.savez_compressed(os.path.join(HERE, "sdtree_golden.npz"), **out)
    print("wrote sdtree_golden.npz:", {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})


if __name__ == "__main__":
    main()
