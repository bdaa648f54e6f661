"""Parity of the stand-alone SD-tree entry points against the CPU oracle on SURVEY.md 8(d)'s synthetic inputs AT THEIR
STATED SIZES -- the inputs bench.py times (`kernels_synthetic`, `roofline.s1_* / s2_* / s3_*`), made by the same
generator (practical_path_guiding_lab_amd.workload, compared with the oracle's own streams in tests/test_host_logic.py):

  S1 "balanced"  KD complete to depth 12 (4096 leaves), every leaf a complete quadtree of depth 5 (5.59 M nodes);
                 2^22 queries: pg_get_leaf_node_index, pg_pdf, pg_sample (directions, pdfs, sampler states),
                 pg_guide_bounce = the reference's three calls (path_guiding_integrator.py:244, 301, 307)
  S2 "skewed"    six splat + refine iterations of 2^19 ... 2^24 clustered records (33.5 M in all) on the device and on
                 the oracle: the refined trees are equal key by key; then the same 2^22 queries
  S3 "splat"     the 2^24-record stream replayed into the S2 topology: every accumulator of every node, exactly

The oracle walks these single-threaded at 1-10 M units/s: the module takes about two minutes.  Bit-exact throughout
(integer and index work; fp32 under DESIGN.md 4's arithmetic contract)."""
import numpy as np
import pytest

from oracle import pg_oracle as po

pytestmark = pytest.mark.gpu


def _np(t):
    return np.ascontiguousarray(t.detach().cpu().numpy())


def _same_tree(a, b):
    assert set(a) == set(b)
    for k in a:
        np.testing.assert_array_equal(np.asarray(a[k]).astype(np.float64), np.asarray(b[k]).astype(np.float64), err_msg=k)


@pytest.fixture(scope="module")
def queries():
    import torch
    from practical_path_guiding_lab_amd import workload as W

    dev = torch.device("cuda", 0)
    return W.s_positions_uniform(W.S_QUERIES, 3, device=dev), W.s_directions_uniform(W.S_QUERIES, 4, device=dev)


def _query_parity(g, o, P, D):
    import torch
    from practical_path_guiding_lab_amd.sdtree import PCG32Sampler

    n = P.shape[1]
    p, d = _np(P), _np(D)
    np.testing.assert_array_equal(_np(g.getLeafNodeIndex(P)).astype(np.uint32), o.get_leaf_node_index(p))
    np.testing.assert_array_equal(_np(g.pdf(P, D)).view(np.uint32), o.pdf(p, d).view(np.uint32))
    smp = PCG32Sampler(g, n, seed=0)  # (PCG32 stream = lane id, seed 0: SURVEY 8d)
    st, inc = po.rng_seed(n, 0)
    d_g, pdf_g = g.sample(P, smp)
    d_o, pdf_o = o.sample(p, st, inc)
    np.testing.assert_array_equal(_np(d_g).view(np.uint32), d_o.view(np.uint32))
    np.testing.assert_array_equal(_np(pdf_g).view(np.uint32), pdf_o.view(np.uint32))
    np.testing.assert_array_equal(_np(smp.state).view(np.uint64), st)
    # the bounce as bench.py times it: NEE pdf for every lane, odd lanes sample, even lanes evaluate
    nee = torch.ones(n, dtype=torch.uint8, device=P.device)
    sel = (torch.arange(n, device=P.device) % 2 + 1).to(torch.uint8)
    dio = D.clone()
    smp2 = PCG32Sampler(g, n, seed=0)
    st2, inc2 = po.rng_seed(n, 0)
    pn_g, pw_g = g.guideBounce(P, D, nee, sel, dio, smp2)
    sel_h = _np(sel)
    ds_o, ps_o = o.sample(p, st2, inc2, (sel_h == 2).astype(np.uint8))
    pb_o = o.pdf(p, d, (sel_h == 1).astype(np.uint8))
    np.testing.assert_array_equal(_np(pn_g).view(np.uint32), o.pdf(p, d).view(np.uint32))
    np.testing.assert_array_equal(_np(pw_g).view(np.uint32), np.where(sel_h == 2, ps_o, pb_o).astype(np.float32).view(np.uint32))
    np.testing.assert_array_equal(_np(dio).view(np.uint32), np.where(sel_h == 2, ds_o, d).astype(np.float32).view(np.uint32))
    np.testing.assert_array_equal(_np(smp2.state).view(np.uint64), st2)


def test_s1_queries_at_full_size(queries):
    from practical_path_guiding_lab_amd import workload as W
    from practical_path_guiding_lab_amd.sdtree import SDTree

    t = W.s1_balanced_tree()
    assert t["kdtree_isLeaf"].sum() == 4096 and t["quadtree_depth"].shape[0] == 4096 * 1365
    g = SDTree(0)
    g.load(t)
    st = g.stats()
    assert st.n_kd_leaves == 4096 and st.mean_kd_leaf_depth == 12.0 and st.mean_quad_leaf_depth == 5.0 and st.jump_bits == 6
    _same_tree(g.export(), t)   # (import -> device layout -> export reproduces the 23 columns)
    o = po.OracleTree()
    o.load(t)
    _query_parity(g, o, *queries)


def test_s2_lifecycle_s3_splat_and_queries_at_full_size(queries):
    import torch
    from practical_path_guiding_lab_amd import workload as W
    from practical_path_guiding_lab_amd.sdtree import SDTree

    dev = torch.device("cuda", 0)
    g = SDTree(0)
    g.setup([W.S_BBOX[0]] * 3, [W.S_BBOX[1]] * 3, 0, 0, 20, 20, True, 0.5)
    pair = po.OracleSDTreePair()
    pair.setup([W.S_BBOX[0]] * 3, [W.S_BBOX[1]] * 3, 20, 20, True)
    rec = None
    for k in range(W.S2_ITERATIONS):
        g.setIteration(k, False)
        rec = W.s2_record_stream(k, device=dev)
        g.addDataPropagate(rec)
        h = {kk: _np(v) for kk, v in rec.items()}
        pair.current.add_data_propagate(h["position"], h["direction"], h["radiance"], h["woPdf"], h["direction_nee"],
                                        h["radiance_nee_lum"])
        g.refineAndPrepare()
        pair.refine_and_prepare(k)
        if k in (2, W.S2_ITERATIONS - 1):
            _same_tree(g.export(), pair.prev.export())
    assert rec["radiance"].shape[0] == W.S2_RECORDS
    st = g.stats()
    assert st.n_kd_leaves > 400 and st.max_quad_depth >= 8   # (434 leaves, quadtrees up to 13 levels deep)
    # S3: the last stream again, into the refined topology; every accumulator of every canonical node
    g.setIteration(W.S2_ITERATIONS, False)
    g.addDataPropagate(rec)
    pair.current.add_data_propagate(h["position"], h["direction"], h["radiance"], h["woPdf"], h["direction_nee"], h["radiance_nee_lum"])
    kd, lo, hi = g.exportAccumulators()
    np.testing.assert_array_equal(kd, pair.current.kd_column("count"))
    np.testing.assert_array_equal(lo, pair.current.quad_column("acc_lo"))
    np.testing.assert_array_equal(hi, pair.current.quad_column("acc_hi"))
    assert int(kd[0]) == W.S2_RECORDS   # every record lies inside the root box and is counted at the root
    del rec
    _query_parity(g, pair.prev, *queries)
