"""Pins the oracle's SD-tree behaviour with the invariants the reference's own self-tests print
(src/quadtree.py:1106-1436, src/kdtree.py:667-835) plus hand-derivable cases.
CPU only."""
import numpy as np
import pytest

import synth
from oracle import pg_oracle as po

BB0, BB1 = [0.0] * 3, [100.0] * 3


def fresh(kd_depth=20, quad_depth=20, nee=True):
    t = po.OracleTree()
    t.setup(BB0, BB1, kd_depth, quad_depth, nee)
    return t


def bbox_nesting_ok_quad(d):
    # quadtree.py:468-509 validateQuadTreeNodeBBox
    nl = ~d["quadtree_isLeaf"]
    for k in (1, 2, 3, 4):
        c = d["quadtree_child_%d_index" % k][nl]
        if not ((d["quadtree_bbox_min"][c] >= d["quadtree_bbox_min"][nl]).all()
                and (d["quadtree_bbox_max"][c] <= d["quadtree_bbox_max"][nl]).all()):
            return False
    return True


def bbox_nesting_ok_kd(d):
    # kdtree.py:361-398 validateTreeNodeBBox
    nl = ~d["kdtree_isLeaf"]
    for k in ("left", "right"):
        c = d["kdtree_child_%s_index" % k][nl]
        if not ((d["kdtree_bbox_min"][c] >= d["kdtree_bbox_min"][nl]).all()
                and (d["kdtree_bbox_max"][c] <= d["kdtree_bbox_max"][nl]).all()):
            return False
    return True


def test_initial_state():
    # kdtree.py:122-124, quadtree.py:355-359
    d = fresh().export()
    assert d["kdtree_depth"].shape == (1,) and d["kdtree_isLeaf"][0]
    assert d["quadtree_depth"].shape == (1,) and d["quadtree_isLeaf"][0]
    np.testing.assert_array_equal(d["quadtree_bbox_min"], [[0, 0]])
    np.testing.assert_array_equal(d["quadtree_bbox_max"], [[1, 1]])
    assert np.isinf(d["quadtree_refinementThreshold"][0]) and d["quadtree_irradiance"][0] == 0
    np.testing.assert_array_equal(d["quadtree_rootNodeIndex"], [0])


def test_three_full_quad_splits_give_85_nodes():
    # quadtree.py:1143-1155
    t = fresh()
    for _ in range(3):
        t.quad_split(t.quad_all_leaves())
    d = t.export()
    assert d["quadtree_depth"].shape[0] == 1 + 4 + 16 + 64 == 85
    assert d["quadtree_isLeaf"].sum() == 64 and d["quadtree_depth"].max() == 3
    assert bbox_nesting_ok_quad(d)
    # quadrants (quadtree.py:153-175): child1 = [mid,max], child2 = x<=mid & y>=mid, child3 = [min,mid], child4
    r = 0
    c = [d["quadtree_child_%d_index" % k][r] for k in (1, 2, 3, 4)]
    np.testing.assert_array_equal(d["quadtree_bbox_min"][c], [[0.5, 0.5], [0, 0.5], [0, 0], [0.5, 0]])
    np.testing.assert_array_equal(d["quadtree_bbox_max"][c], [[1, 1], [0.5, 1], [0.5, 0.5], [1, 0.5]])


def test_two_full_kd_splits_give_7_nodes_4_leaves_and_clone_trees():
    # kdtree.py:708-728; each split keeps one quadtree and clones one (kdtree.py:316-323)
    t = fresh()
    t.quad_split(t.quad_all_leaves())
    for _ in range(2):
        t.kd_split(t.kd_all_leaves())
    d = t.export()
    assert d["kdtree_depth"].shape[0] == 7 and d["kdtree_isLeaf"].sum() == 4
    assert bbox_nesting_ok_kd(d)
    assert d["quadtree_rootNodeIndex"].shape[0] == 4
    assert d["quadtree_depth"].shape[0] == 4 * 5
    leaves = np.nonzero(d["kdtree_isLeaf"])[0]
    assert sorted(d["kdtree_quadTreeRootIndex"][leaves].tolist()) == [0, 1, 2, 3]
    # split axis = depth % 3, children meet at the fp32 midpoint (kdtree.py:270-297)
    np.testing.assert_array_equal(d["kdtree_bbox_max"][1], [50, 100, 100])
    np.testing.assert_array_equal(d["kdtree_bbox_min"][2], [50, 0, 0])
    # left child inherits the parent's tree, a stale non-leaf entry equals its leftmost leaf's (A13)
    assert d["kdtree_quadTreeRootIndex"][0] == 0 and d["kdtree_quadTreeRootIndex"][1] == 0
    # round numbering: children of the i-th split node at old+2i, old+2i+1 (kdtree.py:243-245)
    np.testing.assert_array_equal(d["kdtree_child_left_index"][:3], [1, 3, 5])
    np.testing.assert_array_equal(d["kdtree_child_right_index"][:3], [2, 4, 6])


def test_kd_descent_tie_goes_right_and_outside_goes_to_node0():
    t = fresh()
    t.kd_split(t.kd_all_leaves())
    p = np.array([[49.999996, 50.0, 50.000004, -1.0, np.nan], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1]], np.float32)
    np.testing.assert_array_equal(t.get_leaf_node_index(p), [1, 2, 2, 0, 0])
    np.testing.assert_array_equal(t.get_leaf_node_index(p, active=[0, 0, 0, 0, 0]), [0, 0, 0, 0, 0])


def test_flux_and_count_conservation_after_splat():
    # quadtree.py:1208-1218, kdtree.py:745-748, 769-772
    t = fresh()
    for _ in range(2):
        t.quad_split(t.quad_all_leaves())
    for _ in range(3):
        t.kd_split(t.kd_all_leaves())
    m = 20000
    rec = synth.records(m, 11, BB0, BB1)
    rec["position"][:, :10] = -5.0  # ten records outside the bbox
    synth.splat(t, rec)
    cnt = t.kd_column("count")
    leaf = t.kd_column("isLeaf").astype(bool)
    assert cnt[0] == m - 10 and cnt[leaf].sum() == m - 10
    lo, hi = t.quad_column("acc_lo"), t.quad_column("acc_hi")
    acc = np.array([(int(h) << 64) + int(l) for l, h in zip(lo, hi)], dtype=object)
    qleaf = t.quad_column("isLeaf").astype(bool)
    roots = t.quad_column("rootNodeIndex")
    assert acc[roots].sum() == acc[qleaf].sum()
    # every inner node equals the exact sum of its children
    for k in np.nonzero(~qleaf)[0]:
        assert acc[k] == sum(acc[t.quad_column("child_%d_index" % j)[k]] for j in (1, 2, 3, 4))
    # total = sum of quantised path + NEE weights of all records (out-of-bbox ones land in tree 0)
    w = (rec["radiance"] / rec["woPdf"]).astype(np.float32)
    wn = (rec["radiance_nee_lum"] / rec["woPdf"]).astype(np.float32)
    ql, qh = po.quantize(np.concatenate([w, wn]))
    total = sum((int(h) << 64) + int(l) for l, h in zip(ql, qh))
    assert acc[roots].sum() == total
    t.finalize_accumulators()
    irr = t.quad_column("irradiance")
    assert abs(irr[roots].astype(np.float64).sum() - float(total) / 2.0 ** 40) < 1e-6 * float(total) / 2.0 ** 40
    vc = t.kd_column("vertCount")
    assert vc[0] == m - 10


def test_store_nee_flag_controls_second_splat():
    a, b = fresh(nee=True), fresh(nee=False)
    rec = synth.records(5000, 5, BB0, BB1)
    synth.splat(a, rec)
    synth.splat(b, rec)
    ta = (int(a.quad_column("acc_hi")[0]) << 64) + int(a.quad_column("acc_lo")[0])
    tb = (int(b.quad_column("acc_hi")[0]) << 64) + int(b.quad_column("acc_lo")[0])
    assert ta > tb > 0


def test_wopdf_nonpositive_contributes_zero_weight_but_counts():
    # quadtree.py:451: w = select(woPdf > 0, radiance / woPdf, 0)
    t = fresh()
    rec = synth.records(100, 5, BB0, BB1)
    rec["woPdf"][:] = 0.0
    rec["woPdf"][50:] = -1.0
    synth.splat(t, rec)
    assert t.kd_column("count")[0] == 100
    assert t.quad_column("acc_lo")[0] == 0 and t.quad_column("acc_hi")[0] == 0


def test_count_saturates_like_fp32_increment():
    # kdtree.py:199 adds 1.0f with a float atomic: exact to 2^24 and stuck there afterwards
    t = fresh()
    big = {"kdtree_vertCount": None}
    rec = synth.records(1000, 1, BB0, BB1, skew=False)
    synth.splat(t, rec)
    t.finalize_accumulators()
    assert t.kd_column("vertCount")[0] == 1000.0
    f = np.float32(16777216.0)
    assert f + np.float32(1.0) == f  # the property the saturation rule encodes


def test_reset_zeroes_values_keeps_structure():
    # quadtree.py:1306-1311
    pair = synth.build_skewed(1 << 13, 3)
    cur = pair.current.export()
    prv = pair.prev.export()
    assert (cur["quadtree_irradiance"] == 0).all() and (cur["kdtree_vertCount"] == 0).all()
    assert (pair.current.quad_column("acc_lo") == 0).all()
    for k in cur:
        if k in ("quadtree_irradiance", "kdtree_vertCount"):
            continue
        np.testing.assert_array_equal(cur[k], prv[k], err_msg=k)
    assert (prv["quadtree_irradiance"][prv["quadtree_rootNodeIndex"]] > 0).any()


def test_canonical_layout_after_clean():
    # SURVEY A8 (quadtree.py:695-828, 844-851)
    pair = synth.build_skewed(1 << 14, 4)
    d = pair.prev.export()
    R = d["quadtree_rootNodeIndex"].shape[0]
    np.testing.assert_array_equal(d["quadtree_rootNodeIndex"], np.arange(R))
    leaf = d["quadtree_isLeaf"]
    c1 = d["quadtree_child_1_index"]
    for k in (2, 3, 4):
        np.testing.assert_array_equal(d["quadtree_child_%d_index" % k][~leaf], c1[~leaf] + (k - 1))
        assert (d["quadtree_child_%d_index" % k][leaf] == 0).all()
    # level order over the whole forest: first-child pointers of non-leaf nodes are increasing in node order
    assert (np.diff(c1[~leaf].astype(np.int64)) == 4).all()
    assert c1[~leaf][0] == R
    assert (np.diff(d["quadtree_depth"].astype(np.int64)) >= 0).all()
    # every node is reachable exactly once
    seen = np.zeros(leaf.shape[0], np.int32)
    seen[:R] += 1
    for k in (1, 2, 3, 4):
        np.add.at(seen, d["quadtree_child_%d_index" % k][~leaf], 1)
    assert (seen == 1).all()
    assert bbox_nesting_ok_quad(d) and bbox_nesting_ok_kd(d)
    # number of trees == number of KD leaves, bijection (A13)
    kl = d["kdtree_isLeaf"]
    assert kl.sum() == R
    assert sorted(d["kdtree_quadTreeRootIndex"][kl].tolist()) == list(range(R))


def test_refine_postconditions():
    # kdtree.py:341-358, quadtree.py:563-637 (strict < and >)
    pair = synth.build_skewed(1 << 15, 4)
    d = pair.prev.export()
    it = 3
    thr = np.float32(12000.0 * np.sqrt(2.0 ** it))
    kl = d["kdtree_isLeaf"]
    ok = (d["kdtree_vertCount"][kl] <= thr) | (d["kdtree_depth"][kl] >= 20)
    assert ok.all()
    leaf = d["quadtree_isLeaf"]
    irr, th, dep = d["quadtree_irradiance"], d["quadtree_refinementThreshold"], d["quadtree_depth"]
    assert ((irr[leaf] <= th[leaf]) | (dep[leaf] >= 20)).all()
    assert (irr[~leaf] >= th[~leaf]).all()
    # per-tree threshold = root irradiance / 100 (fp32), identical on every node of the tree
    roots = d["quadtree_rootNodeIndex"]
    np.testing.assert_array_equal(th[roots], (irr[roots] / np.float32(100)).astype(np.float32))
    for k in (1, 2, 3, 4):
        c = d["quadtree_child_%d_index" % k][~leaf]
        np.testing.assert_array_equal(th[c], th[~leaf])
    # fresh children carry a quarter of the parent (quadtree.py:133-138); others sum exactly to <= parent
    assert d["quadtree_depth"].max() <= 20


def test_first_refine_of_single_leaf_tree_is_depth4_complete():
    # root irr v, thr = v/100: a leaf splits while v/4^j > v/100 -> j = 4 levels: 1+4+16+64+256 nodes
    t = po.OracleSDTreePair()
    t.setup(BB0, BB1, 20, 20, True)
    rec = synth.records(1000, 3, BB0, BB1)
    synth.splat(t.current, rec)
    t.refine_and_prepare(0)
    d = t.prev.export()
    assert d["kdtree_depth"].shape[0] == 1
    assert d["quadtree_depth"].shape[0] == 341 and d["quadtree_isLeaf"].sum() == 256
    root = d["quadtree_irradiance"][0]
    np.testing.assert_array_equal(d["quadtree_irradiance"][d["quadtree_isLeaf"]], np.float32(root / 256))


def test_kd_split_halves_counts_and_splits_uniformly():
    # kdtree.py:261-264: children get vertCount/2, so one over-full leaf becomes a complete subtree
    t = po.OracleSDTreePair()
    t.setup(BB0, BB1, 20, 20, True)
    m = 100000  # > 12000 * 2^3 = 96000, <= 12000 * 2^4 -> 4 levels
    rec = synth.records(m, 9, BB0, BB1)
    synth.splat(t.current, rec)
    t.refine_and_prepare(0)
    d = t.prev.export()
    assert d["kdtree_depth"].shape[0] == 31 and d["kdtree_isLeaf"].sum() == 16
    np.testing.assert_array_equal(d["kdtree_vertCount"][d["kdtree_isLeaf"]], np.float32(m / 16))
    assert d["quadtree_rootNodeIndex"].shape[0] == 16
    # every clone is structurally identical to the original
    n_per = d["quadtree_depth"].shape[0] // 16
    assert d["quadtree_depth"].shape[0] == 16 * n_per


def test_sample_returns_its_own_pdf_and_unit_dirs():
    pair = synth.build_skewed(1 << 14, 4)
    n = 20000
    p = synth.positions_uniform(n, 5, BB0, BB1)
    st, inc = po.rng_seed(n, 0)
    d, pdf = pair.prev.sample(p, st, inc)
    assert np.abs(np.linalg.norm(d.astype(np.float64), axis=0) - 1).max() < 1e-6
    np.testing.assert_array_equal(pdf, pair.prev.pdf(p, d))
    assert (pdf > 0).all() and np.isfinite(pdf).all()
    # inactive lanes: dir = canonicalToDir(0,0) = (0,0,-1), pdf = 1, RNG untouched (quadtree.py:940-946, 1011-1020)
    st2, inc2 = po.rng_seed(n, 0)
    act = np.zeros(n, np.uint8)
    act[::2] = 1
    d2, pdf2 = pair.prev.sample(p, st2, inc2, active=act)
    np.testing.assert_array_equal(d2[:, ::2], d[:, ::2])
    np.testing.assert_array_equal(d2[:, 1::2], np.tile(np.array([[0], [0], [-1]], np.float32), (1, n // 2)))
    assert (pdf2[1::2] == 1).all()
    s0, _ = po.rng_seed(n, 0)
    np.testing.assert_array_equal(st2[1::2], s0[1::2])
    assert (st2[::2] != s0[::2]).all()


def test_rng_draws_three_uniforms_per_visited_node():
    # next_2d + next_1d per visited node, the leaf included
    t = fresh()
    for _ in range(2):
        t.quad_split(t.quad_all_leaves())
    d = t.export()
    d["quadtree_irradiance"] = np.ones(21, np.float32)
    t.load(d)
    n = 64
    st, inc = po.rng_seed(n, 3)
    st0 = st.copy()
    t.sample(synth.positions_uniform(n, 1, BB0, BB1), st, inc)
    ref = st0.copy()
    for _ in range(9):  # 3 nodes visited (depth 0,1,2) x 3 draws
        po.rng_next_f32(ref, inc)
    np.testing.assert_array_equal(st, ref)


def test_pdf_integrates_to_one_and_matches_sampling_density():
    pair = synth.build_skewed(1 << 14, 4)
    tree = pair.prev
    n = 200000
    # all queries at one position -> one quadtree
    p = np.tile(np.array([[50.5], [50.5], [50.5]], np.float32), (1, n))
    e = tree.export()
    tid = e["kdtree_quadTreeRootIndex"][tree.get_leaf_node_index(p[:, :1])[0]]
    # exact quadrature: the pdf is piecewise constant on the leaf cells (area 4^-depth of the unit square,
    # which maps area-preservingly onto the sphere: d(omega) = 4*pi dx dy)
    stack, cells = [int(e["quadtree_rootNodeIndex"][tid])], []
    while stack:
        k = stack.pop()
        if e["quadtree_isLeaf"][k]:
            cells.append(k)
        else:
            stack += [int(e["quadtree_child_%d_index" % j][k]) for j in (1, 2, 3, 4)]
    cells = np.array(cells)
    ctr = ((e["quadtree_bbox_min"][cells] + e["quadtree_bbox_max"][cells]) / 2).T.astype(np.float32)
    area = (e["quadtree_bbox_max"][cells, 0] - e["quadtree_bbox_min"][cells, 0]).astype(np.float64) ** 2
    pdf_c = tree.pdf_quadtree(np.full(cells.shape[0], tid, np.uint32), po.canonical_to_dir(ctr)).astype(np.float64)
    mass_leaf = pdf_c * 4 * np.pi * area
    assert abs(mass_leaf.sum() - 1.0) < 1e-4
    # histogram of sampled canonical positions on a 4x4 grid vs. the exact mass of each grid cell
    st, inc = po.rng_seed(n, 9)
    ds, _ = tree.sample(p, st, inc)
    cs = po.dir_to_canonical(ds)
    cell_s = np.minimum((cs[0] * 4).astype(int), 3) * 4 + np.minimum((cs[1] * 4).astype(int), 3)
    freq = np.bincount(cell_s, minlength=16) / n
    mass = np.zeros(16)
    for (cx, cy), a, m in zip(ctr.T, area, mass_leaf):
        if a <= 1.0 / 16 + 1e-12:
            mass[min(int(cx * 4), 3) * 4 + min(int(cy * 4), 3)] += m
        else:  # a leaf bigger than a grid cell spreads uniformly
            s = int(round(np.sqrt(a) * 4))
            x0, y0 = int((cx - np.sqrt(a) / 2) * 4 + 0.5), int((cy - np.sqrt(a) / 2) * 4 + 0.5)
            for ix in range(s):
                for iy in range(s):
                    mass[(x0 + ix) * 4 + (y0 + iy)] += m / (s * s)
    assert np.abs(freq - mass).max() < 0.01


def test_tie_rules_on_cell_boundaries():
    # splat & pdf-descent: highest-numbered containing child; pdf energy: lowest-numbered
    t = fresh()
    t.quad_split(t.quad_all_leaves())
    d = t.export()
    d["quadtree_irradiance"] = np.array([10, 1, 2, 3, 4], np.float32)
    t.load(d)
    inv4pi = np.float32(0.07957747154594766788)
    # canonical points exactly on the boundaries, expressed through the root ids (quadtree-level API)
    pts = {  # (x, y) -> (energy child [1-based], next child [1-based])
        (0.5, 0.75): (1, 2), (0.5, 0.25): (3, 4), (0.75, 0.5): (1, 4), (0.25, 0.5): (2, 3), (0.5, 0.5): (1, 4),
    }
    for (x, y), (ce, cn) in pts.items():
        # build a direction whose canonical image is exactly (x, y): use the splat path for `next`
        t2 = fresh()
        t2.quad_split(t2.quad_all_leaves())
        t2.add_data_propagate(np.array([[1.0], [1.0], [1.0]], np.float32), np.array([[x], [y]], np.float32),
                              np.array([1.0], np.float32), np.array([1.0], np.float32),
                              np.array([[x], [y]], np.float32), np.array([0.0], np.float32))
        lo = t2.quad_column("acc_lo")
        hit = np.nonzero(lo[1:] != 0)[0] + 1
        assert hit.tolist() == [cn], ((x, y), hit)
    # pdf energy pick at (0.5, y>0.5): canonical x = 0.5 <=> phi = pi <=> dir = (-s, ~0, c); atan2 gives exactly pi
    dirs = np.array([[-1.0], [0.0], [0.0]], np.float32)  # canonical (0.5, 0.5): energy child 1, leaf next
    pdf = t.pdf_quadtree(np.zeros(1, np.uint32), dirs)
    assert pdf[0] == np.float32(np.float32(np.float32(4.0) * np.float32(1.0)) / np.float32(10.0)) * inv4pi


def test_zero_energy_tree_pdf_is_zero_and_sampling_takes_child4():
    # quadtree.py:983-991 (all-zero CDF -> child 4), 1086-1092 (NaN pdf -> 0)
    t = fresh()
    for _ in range(2):
        t.quad_split(t.quad_all_leaves())
    n = 16
    st, inc = po.rng_seed(n, 1)
    p = synth.positions_uniform(n, 2, BB0, BB1)
    d, pdf = t.sample(p, st, inc)
    c = po.dir_to_canonical(d)
    # child 4 twice = x in [0.75,1], y in [0,0.25]
    assert (c[0] >= 0.75 - 1e-6).all() and (c[1] <= 0.25 + 1e-6).all()
    assert (pdf == 0).all()


def test_out_of_bbox_uses_tree_zero():
    pair = synth.build_skewed(1 << 14, 4)
    n = 100
    inside = np.tile(np.array([[1e-3], [1e-3], [1e-3]], np.float32), (1, n))
    outside = np.tile(np.array([[-3.0], [50.0], [50.0]], np.float32), (1, n))
    d = synth.directions_uniform(n, 4)
    e = pair.prev.export()
    leaf0 = pair.prev.get_leaf_node_index(inside)[0]
    assert e["kdtree_quadTreeRootIndex"][leaf0] == 0  # leftmost leaf keeps tree 0
    np.testing.assert_array_equal(pair.prev.pdf(outside, d), pair.prev.pdf(inside, d))


def test_process_records_known_answers(oracle):
    # path_guiding_integrator.py:434-500, one ray with max_depth 4
    R, D = 1, 4
    S = R * D
    L = np.array([[3.0], [3.0], [3.0]], np.float32)
    rec = {
        "active": np.array([1, 1, 1, 0], np.uint8),
        "position": np.arange(3 * S, dtype=np.float32).reshape(3, S),
        "direction": np.full((2, S), 0.25, np.float32),
        "bsdf": np.full((3, S), 0.5, np.float32),
        "throughputBsdf": np.full((3, S), 2.0, np.float32),
        "throughputRadiance": np.array([[1.0, 3.0, 1.0, 0.0]] * 3, np.float32),
        "radiance_nee": np.zeros((3, S), np.float32),
        "direction_nee": np.full((2, S), 0.75, np.float32),
        "woPdf": np.array([0.5, 0.5, 0.0, 0.5], np.float32),
    }
    out = oracle.process_records(R, D, L, rec)
    # slot0: out=(3-1)/2=1, in=1/0.5=2, lum(2,2,2)=2*(sum of weights); kept
    # slot1: radiance 0 and nee 0 -> dropped; slot2: woPdf 0 -> dropped; slot3: inactive
    assert out["radiance"].shape == (1,)
    w = np.float32(2.0)
    lum = np.float32(np.float32(np.float32(w * np.float32(0.212671)) + np.float32(w * np.float32(0.715160))) + np.float32(w * np.float32(0.072169)))
    assert out["radiance"][0] == lum
    np.testing.assert_array_equal(out["position"][:, 0], [0, 4, 8])
    # NaN scrubbing: 0/0 -> 0 (path_guiding_integrator.py:444, 449)
    rec["throughputBsdf"][:, 0] = 0.0
    rec["throughputRadiance"][:, 0] = 3.0
    rec["radiance_nee"][:, 0] = 1.0
    out = oracle.process_records(R, D, L, rec)
    assert out["radiance"].tolist() == [0.0] and out["radiance_nee_lum"][0] > 0.99


def test_threaded_stand_alone_entry_points_equal_the_single_threaded_ones():
    """bench.py's CPU columns for S1 / S2 / S3 run pgo_get_leaf_node_index, pgo_pdf, pgo_sample and
    pgo_add_data_propagate on all host cores (pgo_set_threads): lanes are independent and the splat adds exact
    integers (a 128-bit add made of two fetch-and-adds with the carry taken from the first one's old value), so every
    output, every sampler state and every accumulator -- negative, huge and NaN weights included -- is the
    single-threaded one, bit for bit."""
    import synth
    from oracle import pg_oracle as po

    pair = synth.build_skewed(1 << 12, 4)
    tree = pair.prev
    n = 1 << 15
    p = synth.positions_uniform(n, 5, [0.0] * 3, [100.0] * 3)
    p[:, :7] = np.float32(1e9)     # outside the box
    d = synth.directions_uniform(n, 6)
    rec = synth.records(1 << 16, 99, [0.0] * 3, [100.0] * 3)
    rec["radiance"][::7] *= np.float32(-1.0)          # negative weights borrow across the 64-bit halves
    rec["radiance"][3::11] = np.float32(3e38)         # clamped to 2^48: the carries of the low half
    rec["radiance"][5::13] = np.float32("nan")
    rec["woPdf"][::17] = np.float32(1e-30)
    out = {}
    try:
        for threads in (1, 7):
            assert po.set_threads(threads) == threads
            st, inc = po.rng_seed(n, 0)
            dirs, pdfs = tree.sample(p, st, inc)
            cur = po.OracleTree()
            cur.copy_from(pair.current)
            for _ in range(3):
                synth.splat(cur, rec)
            out[threads] = (tree.get_leaf_node_index(p), tree.pdf(p, d), dirs, pdfs, st.copy(),
                            cur.kd_column("count"), cur.quad_column("acc_lo"), cur.quad_column("acc_hi"))
    finally:
        po.set_threads(1)
    for a, b in zip(out[1], out[7]):
        assert a.dtype == b.dtype and a.tobytes() == b.tobytes()
    assert out[1][5][0] == 3 * (1 << 16) and int(np.abs(out[1][7]).max()) > 0   # the high halves are in use


def test_the_scalar_oracle_equals_a_wavefront_restatement_of_the_reference():
    """oracle/pg_oracle.c walks lane by lane; the reference is a masked wavefront program (masked gathers yield 0, masked
    scatters overwrite in statement order).  tests/wavefront_model.py restates getLeafNodeIndex, pdfQuadTree, sampleQuadTree
    and addIrradiancePropagate in THAT form on the 23-key columns: two independently structured programs must agree bit for
    bit -- leaf indices (ties on split planes, points outside the box, NaN), pdfs (ties on cell boundaries take the first
    child's energy and walk into the last), sampled directions and sampler states (three draws per visited node, the leaf
    included), and the nodes a record adds to."""
    import synth
    import wavefront_model as wm
    from oracle import pg_oracle as po

    pair = synth.build_skewed(1 << 12, 4)
    tree = pair.prev
    t = tree.export()
    n = 6000
    p = synth.positions_uniform(n, 21, [0.0] * 3, [100.0] * 3)
    # points ON split planes (ties go right), outside the box, NaN
    kd_lo, kd_hi = t["kdtree_bbox_min"], t["kdtree_bbox_max"]
    inner = np.nonzero(~t["kdtree_isLeaf"])[0]
    for j, node in enumerate(inner[:40]):
        right = t["kdtree_child_right_index"][node]
        axis = int(t["kdtree_depth"][node]) % 3
        q = (0.5 * (kd_lo[right].astype(np.float64) + kd_hi[right].astype(np.float64))).astype(np.float32)
        q[axis] = kd_lo[right][axis]                      # the plane the two children share
        p[:, j] = q
    p[:, 100] = np.float32(1e9)
    p[0, 101] = np.float32("nan")
    np.testing.assert_array_equal(wm.kd_get_leaf_node_index(t, p.T.copy()), tree.get_leaf_node_index(p))
    leaf = tree.get_leaf_node_index(p)
    roots = t["kdtree_quadTreeRootIndex"][leaf].astype(np.uint32)
    # directions: random, and canonical positions exactly on cell boundaries of every depth
    d = synth.directions_uniform(n, 22)
    c = synth.canonical_uniform(n, 23)
    c[:, :2000] = np.floor(c[:, :2000] * np.float32(32)) / np.float32(32)     # multiples of 1/32: on the boundaries of levels <= 5
    d[:, :3000] = po.canonical_to_dir(c)[:, :3000]
    d[2, 3001] = np.float32("nan")
    pdf_o = tree.pdf_quadtree(roots, d)
    pdf_w = wm.quad_pdf(t, roots, d.T.copy())
    np.testing.assert_array_equal(pdf_w.view(np.uint32), pdf_o.view(np.uint32))
    assert (pdf_o > 0).sum() > n // 2
    st_o, inc = po.rng_seed(n, 7)
    st_w = st_o.copy()
    dir_o = tree.sample_quadtree(roots, st_o, inc)
    dir_w, _ = wm.quad_sample(t, roots, st_w, inc)
    np.testing.assert_array_equal(dir_w.T.copy().view(np.uint32), dir_o.view(np.uint32))
    np.testing.assert_array_equal(st_w, st_o)
    # the nodes a record adds to (quadtree.py:398-441): every lane's path, node for node, against the oracle's accumulators
    m = 3000
    cur = po.OracleTree()
    cur.copy_from(tree)
    cur.reset()
    pos2 = c[:, :m].T.copy()
    pos2[5] = [np.float32(1.5), np.float32(0.2)]          # outside the unit square: adds to nothing
    w = np.ones(m, np.float32)
    zero2 = np.full((2, m), np.float32(2.0))             # (the NEE direction outside the square: no second add)
    cur.add_data_propagate(p[:, :m], np.ascontiguousarray(pos2.T), w, w, zero2, np.zeros(m, np.float32))
    got = cur.quad_column("acc_lo")
    exp = np.zeros_like(got)
    q1 = int(po.quantize(np.ones(1, np.float32))[0][0])
    for lanes, nodes in wm.quad_add_propagate(t, roots[:m], pos2):
        np.add.at(exp, nodes, np.uint64(q1))
    np.testing.assert_array_equal(got, exp)
    assert exp.sum() > 0 and (cur.quad_column("acc_hi") == 0).all()
