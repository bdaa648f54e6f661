"""End-to-end parity of the device integrator loop (pg_render_pass) with the CPU restatement of
PathGuidingIntegrator.sample() (oracle/pg_oracle_render.c) on the cornell-box scene: for the same
sampler seeds the per-lane radiance, the valid flags, the per-pixel sums, the splatted accumulators
and the refined SD-trees are bit-identical over several training iterations (guiding switches on at
iteration 2, path_guiding_integrator.py:223)."""
import numpy as np
import pytest

from oracle import pg_oracle as po

pytestmark = pytest.mark.gpu


def _same_tree(a, b):
    assert set(a) == set(b)
    for k in a:
        np.testing.assert_array_equal(np.asarray(a[k]).astype(np.float64), np.asarray(b[k]).astype(np.float64), err_msg=k)


@pytest.mark.parametrize("res,max_depth,rr_depth,nee,boxes", [(48, 8, 8, True, True), (32, 12, 3, False, False)])
def test_cornell_lifecycle_bit_exact(res, max_depth, rr_depth, nee, boxes):
    import torch
    from practical_path_guiding_lab_amd.integrator import PathGuidingIntegrator
    from practical_path_guiding_lab_amd.render import IndependentSampler, WavefrontScene
    from practical_path_guiding_lab_amd.scene import cornell_box

    # boxes: the two cubes as box primitives + a material table, or as six quads each without one
    sc = cornell_box(res, res, max_depth, rr_depth, boxes=boxes)
    assert sc.boxes.shape[0] == (2 if boxes else 0) and sc.quads.shape[0] == (6 if boxes else 18)
    bmin, bmax = sc.bbox_min - np.float32(1e-4), sc.bbox_max + np.float32(1e-4)  # main.py:55-59
    npix = res * res
    o = po.OracleSDTreePair()
    o.setup(bmin, bmax, 20, 20, nee)
    o_sumL = np.zeros((3, npix), np.float32)
    o_sumL2 = np.zeros((3, npix), np.float32)
    g = PathGuidingIntegrator({"max_depth": max_depth, "rr_depth": rr_depth})
    g.setup(npix, bmin, bmax, sdTreeMaxDepth=20, quadTreeMaxDepth=20, isStoreNEERadiance=nee, bsdfSamplingFraction=0.5)
    ws = WavefrontScene(sc)
    cumm = 0
    for k in range(4):
        final = k == 3
        g.setIteration(k, final)
        # mixed pass sizes: 1 spp (the reference's training passes) and a batched pass
        for spp in ([1, 1, 2] if not final else [4]):
            seed = 1000 + cumm
            Lo, vo = po.render_pass(o, sc, sc.camera, max_depth, rr_depth, k, final, seed, spp, nee, 0.5,
                                    o_sumL, o_sumL2)
            Lg, vg, _ = g.sample(ws, IndependentSampler(spp, seed))
            np.testing.assert_array_equal(Lg.cpu().numpy().view(np.uint32), Lo.view(np.uint32))
            np.testing.assert_array_equal(vg.cpu().numpy(), vo)
            cumm += spp
        np.testing.assert_array_equal(g.sumL.cpu().numpy().view(np.uint32), o_sumL.view(np.uint32))
        np.testing.assert_array_equal(g.sumL2.cpu().numpy().view(np.uint32), o_sumL2.view(np.uint32))
        kd, lo, hi = g.sdTree.exportAccumulators()
        np.testing.assert_array_equal(kd, o.current.kd_column("count"))
        np.testing.assert_array_equal(lo, o.current.quad_column("acc_lo"))
        np.testing.assert_array_equal(hi, o.current.quad_column("acc_hi"))
        if not final:
            assert kd[0] > 0
            o.refine_and_prepare(k)
            g.refineAndPrepareSDTreeForNextIteration()
            _same_tree(o.prev.export(), g.sdTree.export())
        else:
            assert not kd.any()  # nothing is recorded in a final iteration (:320, 388)
    # metric plumbing (path_guiding_integrator.py:503-550) on identical sums
    gt = torch.full((3, npix), 0.1, device="cuda")
    mse = g.computeMSE(cumm, gt)
    L = o_sumL.astype(np.float64) / cumm
    d2 = (L - 0.1) ** 2
    ref = np.minimum(0.212671 * d2[0] + 0.715160 * d2[1] + 0.072169 * d2[2], 1e4).mean()
    assert abs(mse - ref) <= 1e-5 * max(ref, 1e-12) + 1e-9
    assert g.computeVariance(cumm) >= 0.0


def mixed_scene(res, max_depth=6, rr_depth=3):
    """Every primitive, emitter kind and BSDF of the substrate in one small scene: a diffuse floor
    and back wall, a rough-conductor plate and cube (two roughnesses), a diffuse and a rough-conductor
    sphere, one quad light and two sphere lights of very different sizes (scenes/veach-mis in small)."""
    from practical_path_guiding_lab_amd import scene as S

    mats = [S.diffuse_material((0.6, 0.55, 0.5)), S.diffuse_material((0.0, 0.0, 0.0)),
            S.roughconductor_material(0.05, (0.200438, 0.924033, 1.10221), (3.91295, 2.45285, 2.14219), (0.3, 0.3, 0.3)),
            S.roughconductor_material(0.25, (0.200438, 0.924033, 1.10221), (3.91295, 2.45285, 2.14219), (0.9, 0.8, 0.7)),
            S.diffuse_material((0.2, 0.5, 0.7))]

    def with_mat(qs, mi):
        for q in qs:
            q[22] = mi
        return qs

    def M(*v):
        return np.array(v, np.float64).reshape(4, 4)

    quads = []
    quads += with_mat(S.rectangle(M(4, 0, 0, 0, 0, 0, 4, 0, 0, -4, 0, 0, 0, 0, 0, 1), mats[0][1:4]), 0)          # floor y=0
    quads += with_mat(S.rectangle(M(4, 0, 0, 0, 0, 4, 0, 4, 0, 0, 1, -4, 0, 0, 0, 1), mats[0][1:4]), 0)          # back wall z=-4
    quads += with_mat(S.rectangle(M(1.5, 0, 0, -1, 0, 0.3, 0.8, 0.6, 0, -1.2, 0.2, 0.5, 0, 0, 0, 1), mats[2][1:4]), 2)  # tilted glossy plate
    quads += with_mat(S.cube(M(0.5, 0.2, 0, 2, 0, 0, -0.6, 0.6, -0.2, 0.5, 0, -1, 0, 0, 0, 1), mats[3][1:4]), 3)    # rough cube
    quads += with_mat(S.rectangle(M(0.4, 0, 0, -2.5, 0, 0, -0.4, 3.5, 0, 0.4, 0, -1, 0, 0, 0, 1), mats[1][1:4], (9, 8, 7)), 1)  # quad light, facing down
    spheres = [S.sphere((0.0, 3.0, 0.5), 0.4, 1, (12.0, 12.0, 12.0)), S.sphere((1.5, 2.5, 1.5), 0.03, 1, (900.0, 700.0, 500.0)),
               S.sphere((-1.8, 0.5, 1.0), 0.5, 4), S.sphere((0.5, 0.45, 1.8), 0.45, 2)]
    boxes = [S.box(M(0.4, 0.1, 0, -2.8, -0.1, 0.5, 0.05, 0.52, 0, -0.05, 0.6, -1.8, 0, 0, 0, 1), 4),   # sheared diffuse box
             S.box(M(0.3, 0, 0.2, 3.0, 0, 0.8, 0, 0.8, -0.2, 0, 0.3, 1.2, 0, 0, 0, 1), 2)]               # glossy pillar
    from practical_path_guiding_lab_amd import mesh as MS
    v, f = MS.icosphere(2)                                                                                 # 320 triangles behind a BVH
    tris = [MS.triangles(v, f, M(0.6, 0, 0, 1.2, 0, 0.5, 0, 1.6, 0, 0, 0.6, -0.5, 0, 0, 0, 1), 3, v),      # glossy ellipsoid, smooth-shaded
            MS.triangles(v[:12], np.array([[0, 11, 5], [0, 5, 1], [3, 9, 4], [3, 4, 2]]), M(0.5, 0, 0, -0.5, 0, 0.5, 0, 2.2, 0, 0, 0.5, 2.0, 0, 0, 0, 1), 4)]  # loose diffuse triangles
    # delta lobes, a one-sided surface and a delta light (scenes/torus): an aluminium mirror, a glass ball and
    # a glass block, a one-sided diffuse card, a directional light next to the area lights
    mats += [S.conductor_material(*S.CONDUCTOR_PRESETS["Al"]), S.dielectric_material(1.5, 1.000277),
             S.diffuse_material((0.8, 0.8, 0.4), twosided=False)]
    quads += with_mat(S.rectangle(M(0.9, 0, 0, 3.0, 0, 0.9, 0.3, 2.0, 0, -0.3, 0.9, -3.2, 0, 0, 0, 1), mats[5][1:4]), 5)      # mirror
    quads += with_mat(S.rectangle(M(0.7, 0, 0, -3.0, 0, 0.7, 0, 2.4, 0, 0, 0.7, -2.0, 0, 0, 0, 1), mats[7][1:4]), 7)          # one-sided card
    spheres.append(S.sphere((-0.4, 0.5, 2.6), 0.5, 6))
    boxes.append(S.box(M(0.5, 0, 0, 2.2, 0, 0.35, 0, 0.36, 0, 0, 0.5, 2.4, 0, 0, 0, 1), 6))
    mats.append(S.roughdielectric_material(0.1, 1.49, 1.000277))                                           # frosted glass (scenes/torus)
    spheres.append(S.sphere((1.0, 0.4, 3.2), 0.4, 8))
    tris.append(MS.triangles(v, f, M(0.35, 0, 0, -1.6, 0, 0.35, 0, 0.36, 0, 0, 0.35, 3.0, 0, 0, 0, 1), 8, v))  # and a smooth-shaded frosted mesh
    mats += [S.roughconductor_material(0.15, *S.CONDUCTOR_PRESETS["Al"], distribution="ggx"),             # the GGX forms (scenes/veach-ajar)
             S.roughdielectric_material(0.2, 1.5, 1.0, "ggx")]
    spheres += [S.sphere((-2.6, 0.3, 2.4), 0.3, 9), S.sphere((2.9, 0.3, 3.3), 0.3, 10)]
    lights = [S.directional_light((0.4, -1.0, -0.3), (1.5, 1.4, 1.2))]
    # textures (scenes/veach-ajar): a small bitmap on a diffuse mesh with texture coordinates, a
    # checkerboard as the specular reflectance of a GGX rough conductor, next to untextured meshes
    rs = np.random.RandomState(7)
    textures = [S.bitmap_texture(rs.randint(0, 256, (5, 7, 3)).astype(np.uint8), (3.0, 2.0, 0.25, -0.5)),
                S.checkerboard_texture((0.9, 0.8, 0.7), (0.1, 0.2, 0.3), (6.0, 4.0, 0.1, 0.2))]
    mats += [S.diffuse_material((0.5, 0.5, 0.5), texture=0),
             S.roughconductor_material(0.2, *S.CONDUCTOR_PRESETS["Al"], distribution="ggx", texture=1)]
    uv = np.stack([np.arctan2(v[:, 1], v[:, 0]) / (2 * np.pi) + 0.5, v[:, 2] * 0.5 + 0.5], axis=1)  # spherical map of the icosphere
    tris.append(MS.triangles(v, f, M(0.45, 0, 0, 0.2, 0, 0.45, 0, 0.46, 0, 0, 0.45, 0.4, 0, 0, 0, 1), 11, None, uv))
    tris.append(MS.triangles(v, f, M(0.4, 0, 0, -0.9, 0, 0.4, 0, 0.41, 0, 0, 0.4, 3.6, 0, 0, 0, 1), 12, v, uv, False))
    cam = S.make_camera(M(-1, 0, 0, 0, 0, 0.94, -0.342, 3.0, 0, -0.342, -0.94, 7.5, 0, 0, 0, 1), 40.0, res, res)
    return S._finish(quads, cam, max_depth, rr_depth, ["q"] * len(quads), spheres, mats, boxes, tris, lights, textures)


@pytest.mark.parametrize("res,nee", [(40, True), (24, False)])
def test_spheres_rough_conductors_and_many_emitters_bit_exact(res, nee):
    """The general kernels (k_bounce<*, true>: spheres, uniform emitter choice with cone-sampled
    sphere lights, Beckmann rough conductors with visible-normal sampling) against the oracle:
    radiance, sums, accumulators and refined trees over a guided lifecycle."""
    _guided_lifecycle_bit_exact(mixed_scene(res), nee)


def test_torus_scene_bit_exact():
    """scenes/torus at 48x36 (23614 triangles behind the BVH, smooth normals, frosted glass, mirrors,
    one-sided diffuse, the directional light, max_depth 30 with roulette from depth 8): the device
    against the oracle over a guided lifecycle."""
    from practical_path_guiding_lab_amd.scene import torus
    _guided_lifecycle_bit_exact(torus(48, 36), True)
    _guided_lifecycle_bit_exact(torus(48, 36), True, stages=1)  # k_wave_shade_a with the SD-tree calls | k_wave_cast | k_wave_shade_b
    _guided_lifecycle_bit_exact(torus(48, 36), True, stages=2)  # ... and the SD-tree calls as k_wave_guide


def test_veach_ajar_scene_bit_exact():
    """scenes/veach-ajar at 64x36 (4482 triangles with texture coordinates, three bitmap textures, the
    checkerboard floor, Beckmann and GGX rough conductors, the smooth-shaded door handle, max_depth 13):
    the split pipeline (feature level 2; the paths left at bounce 4 are finished by k_wave_tail) against
    the oracle over a guided lifecycle."""
    from practical_path_guiding_lab_amd.scene import veach_ajar
    _guided_lifecycle_bit_exact(veach_ajar(64, 36), True)
    _guided_lifecycle_bit_exact(veach_ajar(64, 36), True, stages=1)  # k_wave_shade_a with the SD-tree calls | k_wave_cast | k_wave_shade_b
    _guided_lifecycle_bit_exact(veach_ajar(64, 36), True, stages=2)  # ... and the SD-tree calls as k_wave_guide


@pytest.mark.parametrize("stages", [0, 1, 2])
def test_veach_ajar_deep_split_bounces_bit_exact(stages):
    """veach-ajar at 320x180 with 8 spp per pass: 460 800 paths, more than the tail launch takes over,
    so the deep bounces run through the per-bounce kernels too (the live list, the records of the sorted bounces, the
    workspace planes and the record list at every depth) -- in each of the three forms of pg_render_stages; radiance and
    accumulators against the oracle over three iterations."""
    import torch
    from practical_path_guiding_lab_amd.integrator import PathGuidingIntegrator
    from practical_path_guiding_lab_amd.render import IndependentSampler, WavefrontScene
    from practical_path_guiding_lab_amd.scene import veach_ajar
    po.set_threads(0)
    sc = veach_ajar(320, 180)
    bmin, bmax = sc.bbox_min - np.float32(1e-4), sc.bbox_max + np.float32(1e-4)
    npix = 320 * 180
    o = po.OracleSDTreePair()
    o.setup(bmin, bmax, 20, 20, True)
    g = PathGuidingIntegrator({"max_depth": 13, "rr_depth": 8})
    g.setup(npix, bmin, bmax, 20, 20, True, 0.5)
    ws = WavefrontScene(sc, stages=stages)
    for k in range(3):
        g.setIteration(k, False)
        Lo, vo = po.render_pass(o, sc, sc.camera, 13, 8, k, False, 77 + k, 8, True, 0.5)
        Lg, vg, _ = g.sample(ws, IndependentSampler(8, 77 + k))
        live = g.sdTree.renderLiveCounts(13)
        assert live[5] > 128 * 1024, live  # the split kernels did run deep
        np.testing.assert_array_equal(Lg.cpu().numpy().view(np.uint32), Lo.view(np.uint32))
        np.testing.assert_array_equal(vg.cpu().numpy(), vo)
        kd, lo, hi = g.sdTree.exportAccumulators()
        np.testing.assert_array_equal(kd, o.current.kd_column("count"))
        np.testing.assert_array_equal(lo, o.current.quad_column("acc_lo"))
        np.testing.assert_array_equal(hi, o.current.quad_column("acc_hi"))
        o.refine_and_prepare(k)
        g.refineAndPrepareSDTreeForNextIteration()
        _same_tree(o.prev.export(), g.sdTree.export())
    torch.cuda.synchronize()


@pytest.mark.parametrize("which", ["cornell-box", "veach-mis", "mixed"])
def test_long_paths_finished_by_the_tail_launch_bit_exact(which):
    """max_depth 12-14: beyond depth 4 the few paths left are finished by one k_bounce_tail launch
    (every feature level has its own instantiation); radiance, records, live counts as ever."""
    from practical_path_guiding_lab_amd import scene as S
    sc = {"cornell-box": lambda: S.cornell_box(20, 20, 12, 9), "veach-mis": lambda: S.veach_mis(32, 18, 14, 10),
          "mixed": lambda: mixed_scene(20, max_depth=13, rr_depth=10)}[which]()
    _guided_lifecycle_bit_exact(sc, True)


@pytest.mark.parametrize("which", ["torus 160x120", "veach-ajar 160x90", "mixed 96", "mixed 64 depth 13", "veach-mis 160x90 depth 14",
                                   "cornell-box 96 depth 12"])
def test_lifecycle_parity_at_larger_sizes(which):
    """The guided-lifecycle parity of the tests above (radiance, pixel sums, accumulators and refined trees
    bit for bit against the CPU oracle) at the largest sizes the oracle finishes in seconds, for every
    scene and feature level, including the long-path configurations that end in the tail launch."""
    from practical_path_guiding_lab_amd import scene as S
    po.set_threads(0)
    sc = {"torus 160x120": lambda: S.torus(160, 120), "veach-ajar 160x90": lambda: S.veach_ajar(160, 90),
          "mixed 96": lambda: mixed_scene(96), "mixed 64 depth 13": lambda: mixed_scene(64, max_depth=13, rr_depth=10),
          "veach-mis 160x90 depth 14": lambda: S.veach_mis(160, 90, 14, 10),
          "cornell-box 96 depth 12": lambda: S.cornell_box(96, 96, 12, 9)}[which]()
    _guided_lifecycle_bit_exact(sc, True)


def _guided_lifecycle_bit_exact(sc, nee, bbox=None, stages=0):
    """bbox: the SD-tree's root box when it is not the scene's own (main.py:55-59); stages: pg_render_stages -- a mesh
    scene's bounce as one shading kernel (0), three (1) or four with k_wave_guide on its own (2)."""
    from practical_path_guiding_lab_amd.integrator import PathGuidingIntegrator
    from practical_path_guiding_lab_amd.render import IndependentSampler, WavefrontScene

    D, RR = sc.max_depth, sc.rr_depth
    bmin, bmax = sc.bbox_min - np.float32(1e-4), sc.bbox_max + np.float32(1e-4)
    if bbox is not None:
        bmin, bmax = bbox
    npix = sc.camera.width * sc.camera.height
    o = po.OracleSDTreePair()
    o.setup(bmin, bmax, 20, 20, nee)
    o_sumL = np.zeros((3, npix), np.float32)
    o_sumL2 = np.zeros((3, npix), np.float32)
    g = PathGuidingIntegrator({"max_depth": D, "rr_depth": RR})
    g.setup(npix, bmin, bmax, sdTreeMaxDepth=20, quadTreeMaxDepth=20, isStoreNEERadiance=nee, bsdfSamplingFraction=0.5)
    ws = WavefrontScene(sc, stages=stages)
    cumm = 0
    for k in range(4):
        final = k == 3
        g.setIteration(k, final)
        for spp in ([1, 3] if k == 0 else [2 ** (k + 2)]):
            seed = 5000 + cumm
            Lo, vo = po.render_pass(o, sc, sc.camera, D, RR, k, final, seed, spp, nee, 0.5, o_sumL, o_sumL2)
            Lg, vg, _ = g.sample(ws, IndependentSampler(spp, seed))
            np.testing.assert_array_equal(Lg.cpu().numpy().view(np.uint32), Lo.view(np.uint32))
            np.testing.assert_array_equal(vg.cpu().numpy(), vo)
            cumm += spp
        assert np.isfinite(o_sumL).all() and o_sumL.max() > 0
        np.testing.assert_array_equal(g.sumL.cpu().numpy().view(np.uint32), o_sumL.view(np.uint32))
        kd, lo, hi = g.sdTree.exportAccumulators()
        np.testing.assert_array_equal(kd, o.current.kd_column("count"))
        np.testing.assert_array_equal(lo, o.current.quad_column("acc_lo"))
        np.testing.assert_array_equal(hi, o.current.quad_column("acc_hi"))
        if not final:
            o.refine_and_prepare(k)
            g.refineAndPrepareSDTreeForNextIteration()
            _same_tree(o.prev.export(), g.sdTree.export())


def _block_ratios(img, gt, bh, bw, max_lum=3.0, min_mean=0.01):
    """Ratio of block means img/gt over the blocks that hold no lamp or highlight pixel."""
    lum = gt.mean(axis=2)
    out = []
    for by in range(gt.shape[0] // bh):
        for bx in range(gt.shape[1] // bw):
            sl = (slice(by * bh, (by + 1) * bh), slice(bx * bw, (bx + 1) * bw))
            if lum[sl].max() > max_lum or gt[sl].mean() < min_mean:
                continue
            out.append(img[sl].mean() / gt[sl].mean())
    return np.array(out)


def test_veach_mis_direct_light_matches_the_tungsten_ground_truth():
    """scenes/veach-mis (three sphere lamps of equal power, four Beckmann rough-conductor plates,
    diffuse floor and wall) at 320x180 against the reference's ground truth, an image made by an
    independent renderer (Tungsten).  That image holds direct light only -- it equals this library's
    max_depth 2 render block for block, while every deeper setting is 5-50 % brighter -- so the
    comparison is made at max_depth 2: every 15x20 block without a lamp or highlight pixel agrees
    at main.py's full 1020-spp schedule (8 iterations; measured with this seed: block ratios 0.990 ... 1.037,
    0.23 % off on average, the image mean 0.10 % off).  Guiding is on from iteration 2."""
    import os
    from practical_path_guiding_lab_amd.driver import load_ground_truth, run_guided_render
    from practical_path_guiding_lab_amd.integrator import PathGuidingIntegrator
    from practical_path_guiding_lab_amd.render import WavefrontScene
    from practical_path_guiding_lab_amd.scene import veach_mis

    sc = veach_mis(320, 180, max_depth=2)
    gt_path = os.path.join(os.path.dirname(__file__), "golden", "veach_mis_gt_320x180_f16.npy")
    gt = load_ground_truth(gt_path, 320, 180)
    g = PathGuidingIntegrator({"max_depth": 2, "rr_depth": 8})
    res = run_guided_render(WavefrontScene(sc), g, 1020, initial_seed=3, ground_truth=gt, training_spp_per_pass=4,
                            log=lambda s: None)
    assert res["cumm_spp"] == 1020
    img = res["image"].cpu().numpy().astype(np.float64)
    assert np.isfinite(img).all()
    ratios = _block_ratios(img, np.load(gt_path).astype(np.float64), 15, 20)
    assert ratios.size >= 150
    assert np.abs(ratios - 1).max() < 0.04, (ratios.min(), ratios.max())
    assert np.abs(ratios - 1).mean() < 0.004 and abs(ratios.mean() - 1) < 0.003
    assert np.percentile(np.abs(ratios - 1), 95) < 0.016
    # one more bounce and the image is brighter than the direct-light ground truth everywhere
    sc3 = veach_mis(320, 180, max_depth=3)
    g3 = PathGuidingIntegrator({"max_depth": 3, "rr_depth": 8})
    res3 = run_guided_render(WavefrontScene(sc3), g3, 124, initial_seed=3, training_spp_per_pass=4, log=lambda s: None)
    r3 = _block_ratios(res3["image"].cpu().numpy().astype(np.float64), np.load(gt_path).astype(np.float64), 15, 20)
    assert r3.mean() > 1.04 and r3.min() > 0.99


def test_torus_matches_the_tungsten_image_where_it_converges():
    """scenes/torus at 256x192 through main.py's schedule against the reference's image of it
    (tests/golden/torus_gt_256x192_f16.npy, the linearised TungstenRender.png): the floor in the sun,
    in front of and behind the case, and the case's shadow."""
    import os
    from test_oracle_substrate import TORUS_BLOCKS, TORUS_RTOL, torus_block_means
    from practical_path_guiding_lab_amd.driver import run_guided_render
    from practical_path_guiding_lab_amd.integrator import PathGuidingIntegrator
    from practical_path_guiding_lab_amd.render import WavefrontScene
    from practical_path_guiding_lab_amd.scene import torus

    sc = torus(256, 192)
    sc.rfilter = "box"  # block means of raw pixel estimates
    gt = np.load(os.path.join(os.path.dirname(__file__), "golden", "torus_gt_256x192_f16.npy")).astype(np.float64)
    g = PathGuidingIntegrator({"max_depth": 30, "rr_depth": 8})
    res = run_guided_render(WavefrontScene(sc), g, 124, initial_seed=1, training_spp_per_pass=4, log=lambda s: None)
    img = res["image"].cpu().numpy().astype(np.float64)
    assert img.shape == (192, 256, 3) and np.isfinite(img).all()
    ours, theirs = torus_block_means(img), torus_block_means(gt)
    for k in TORUS_BLOCKS:
        np.testing.assert_allclose(ours[k], theirs[k], rtol=TORUS_RTOL[k], err_msg=k)


def test_guided_render_converges_to_the_ground_truth():
    """main.py's schedule at 256x256 against the reference's ground truth of the same scene
    (tests/golden/cornell_gt_256_f16.npy, from scenes/cornell-box/TungstenRender.exr) over the full
    1020-spp schedule: the MSE metric of path_guiding_integrator.py:530-550 falls as spp grows and the
    mean radiance agrees to 0.5 % (measured -0.50 %: max_depth 8 against the unbounded ground truth)."""
    import os
    from practical_path_guiding_lab_amd.driver import load_ground_truth, run_guided_render
    from practical_path_guiding_lab_amd.integrator import PathGuidingIntegrator
    from practical_path_guiding_lab_amd.render import WavefrontScene
    from practical_path_guiding_lab_amd.scene import cornell_box

    sc = cornell_box(256, 256, 8, 8)
    gt = load_ground_truth(os.path.join(os.path.dirname(__file__), "golden", "cornell_gt_256_f16.npy"), 256, 256)
    g = PathGuidingIntegrator({"max_depth": 8, "rr_depth": 8})
    res = run_guided_render(WavefrontScene(sc), g, 1020, initial_seed=3, ground_truth=gt, training_spp_per_pass=4,
                            log=lambda s: None)
    assert res["cumm_spp"] == 1020
    mse = [r[5] for r in res["records"]["mse_groundTruth_endIter"].rows]
    assert len(mse) == 8 and all(np.isfinite(mse))
    assert mse[-1] < 0.12 * mse[0] and mse[-1] < 4e-4
    img = res["image"].cpu().numpy()
    assert -0.006 * float(gt.mean()) < img.mean() - float(gt.mean()) < 0.0


def test_veach_ajar_agrees_with_the_tungsten_ground_truth():
    """scenes/veach-ajar at 320x180 over main.py's full 1020-spp schedule against the reference's
    ground truth (tests/golden/veach_ajar_gt_320x180_f16.npy from scenes/veach-ajar/TungstenRender.exr),
    outside the rectangle of the three teapots whose meshes the reference mount lacks: the MSE metric
    falls by two orders of magnitude over the iterations; the image mean agrees to 2.5 % and the
    15x20 blocks to 4 % on average, the worst to 15 %.

    The block bound is taken on the mean image of THREE independent runs (seeds 3, 4, 5).  Measured over six seeds
    (tools/ajar_block_spread.py, profiles/r04/ajar_block_spread.txt): what the missing teapots' indirect light and the
    720p ground truth filtered to this film leave is an offset of 3.7 % per block on average and 11.4 % at the worst
    block -- the same blocks, the same sign, whatever the seed; on top of it a block of one run scatters by 1.3 % on
    average and by up to 5.8 % (the scene is lit through the gap of a door), so the worst block of a single run lay
    anywhere between 11.4 % and 15.1 % -- round 3 moved the bound from 0.15 to 0.18 when the run with seed 3 came out
    at 0.1508.  Averaging three runs brings the scatter under 3.4 % and the bound back to 0.15."""
    import os
    from practical_path_guiding_lab_amd.driver import load_ground_truth, run_guided_render
    from practical_path_guiding_lab_amd.integrator import PathGuidingIntegrator
    from practical_path_guiding_lab_amd.render import WavefrontScene
    from practical_path_guiding_lab_amd.scene import veach_ajar, veach_ajar_mask

    gt_path = os.path.join(os.path.dirname(__file__), "golden", "veach_ajar_gt_320x180_f16.npy")
    gtn = np.load(gt_path).astype(np.float64)
    mask = veach_ajar_mask(320, 180)
    imgs = []
    for seed in (3, 4, 5):
        sc = veach_ajar(320, 180)
        assert sc.max_depth == 13 and sc.rfilter == "tent" and len(sc.skipped) == 6
        gt = load_ground_truth(gt_path, 320, 180)
        g = PathGuidingIntegrator({"max_depth": 13, "rr_depth": 8})
        res = run_guided_render(WavefrontScene(sc), g, 1020, initial_seed=seed, ground_truth=gt, training_spp_per_pass=4,
                                log=lambda s: None)
        mse = [r[5] for r in res["records"]["mse_groundTruth_endIter"].rows]
        assert len(mse) == 8 and all(np.isfinite(mse)) and mse[-1] < 0.02 * mse[0]
        img = np.where(mask[..., None], res["image"].cpu().numpy().astype(np.float64), gtn)  # teapot pixels drop out
        assert np.isfinite(img).all()
        assert abs(img[mask].mean() / gtn[mask].mean() - 1) < 0.025
        one = _block_ratios(img, gtn, 15, 20)
        assert one.size >= 150 and abs(one.mean() - 1) < 0.04 and np.abs(one - 1).mean() < 0.05
        imgs.append(img)
    ratios = _block_ratios(np.mean(imgs, axis=0), gtn, 15, 20)
    assert np.abs(ratios - 1).max() < 0.15


def test_tent_film_matches_the_oracle_bit_for_bit():
    """pg_film_tent (hdrfilm + <rfilter type="tent"/>, scenes/cornell-box/scene.xml:27) against
    pgo_film_tent on a ragged film; render() develops the image through it."""
    import ctypes as C
    import torch
    from practical_path_guiding_lab_amd import _native as N
    from practical_path_guiding_lab_amd.integrator import PathGuidingIntegrator
    from practical_path_guiding_lab_amd.render import IndependentSampler, WavefrontScene, render
    from practical_path_guiding_lab_amd.scene import cornell_box

    w, h, spp, seed = 37, 23, 3, 4242
    sc = cornell_box(w, h, 4, 8)
    assert sc.rfilter == "tent"
    g = PathGuidingIntegrator({"max_depth": 4})
    g.setup(w * h, sc.bbox_min - 1e-4, sc.bbox_max + 1e-4, 20, 20, True, 0.5)
    g.setIteration(0, True)
    ws = WavefrontScene(sc)
    L, _, _ = g.sample(ws, IndependentSampler(spp, seed))
    img = torch.empty((3, w * h), dtype=torch.float32, device="cuda")
    t = g.sdTree
    N.check(t._h, t._lib.pg_film_tent(t._h, seed, spp, L.data_ptr(), img.data_ptr(), None))
    exp = po.film_tent(seed, spp, w, h, L.cpu().numpy())
    np.testing.assert_array_equal(img.cpu().numpy().view(np.uint32), exp.view(np.uint32))
    # the same pass through render(): (H, W, 3)
    g2 = PathGuidingIntegrator({"max_depth": 4})
    g2.setup(w * h, sc.bbox_min - 1e-4, sc.bbox_max + 1e-4, 20, 20, True, 0.5)
    g2.setIteration(0, True)
    out = render(WavefrontScene(sc), g2, spp, seed).cpu().numpy()
    np.testing.assert_array_equal(out.view(np.uint32), exp.reshape(3, h, w).transpose(1, 2, 0).view(np.uint32))
    # the gaussian film (hdrfilm's default; scenes/torus/scene.xml:46): 5x5 neighbourhood
    gimg = torch.empty((3, w * h), dtype=torch.float32, device="cuda")
    N.check(t._h, t._lib.pg_film(t._h, 1, seed, spp, L.data_ptr(), gimg.data_ptr(), None))
    gexp = po.film("gaussian", seed, spp, w, h, L.cpu().numpy())
    np.testing.assert_array_equal(gimg.cpu().numpy().view(np.uint32), gexp.view(np.uint32))
    assert not np.array_equal(gexp, exp) and abs(gexp.mean() - exp.mean()) < 0.05 * exp.mean()
    # a box film is the per-pixel mean
    sc.rfilter = "box"
    g3 = PathGuidingIntegrator({"max_depth": 4})
    g3.setup(w * h, sc.bbox_min - 1e-4, sc.bbox_max + 1e-4, 20, 20, True, 0.5)
    g3.setIteration(0, True)
    box = render(WavefrontScene(sc), g3, spp, seed).cpu().numpy()
    np.testing.assert_allclose(box, L.cpu().numpy().reshape(3, h, w, spp).mean(axis=3).transpose(1, 2, 0), rtol=1e-6)
    assert abs(out.mean() - box.mean()) < 0.05 * box.mean()


def test_render_returns_plausible_cornell_image():
    import torch
    from practical_path_guiding_lab_amd.integrator import PathGuidingIntegrator
    from practical_path_guiding_lab_amd.render import WavefrontScene, render
    from practical_path_guiding_lab_amd.scene import cornell_box

    sc = cornell_box(64, 64, 8, 8)
    g = PathGuidingIntegrator({"max_depth": 8})
    g.setup(64 * 64, sc.bbox_min - 1e-4, sc.bbox_max + 1e-4, 20, 20, True, 0.5)
    g.setIteration(0, True)
    img = render(WavefrontScene(sc), g, spp=64, seed=7).cpu().numpy()
    assert img.shape == (64, 64, 3) and np.isfinite(img).all()
    # red wall on the left, green on the right, the light in the top centre
    assert img[32, 2, 0] > 2 * img[32, 2, 1] and img[32, 61, 1] > 2 * img[32, 61, 0]
    assert img[2:6, 28:36].mean() > 3.0 and 0.05 < img.mean() < 0.5


@pytest.mark.parametrize("spp", [5, 24, 28])
def test_both_film_kernels_match_the_oracle(spp):
    """The film is developed by tiles with the samples' film positions staged in LDS (k_film_tiled) where those fit in 64 KB and
    pixel by pixel (k_film) where they do not: 5 samples per pixel take the tiled kernel, 24 take it with more than 48 KB of
    dynamic LDS (the attribute that allows it is set) for the tent filter and the pixel kernel for the gaussian, 28 take the
    pixel kernel for both -- every one against pgo_film on a ragged film, bit for bit."""
    import torch
    from practical_path_guiding_lab_amd import _native as N
    from practical_path_guiding_lab_amd.integrator import PathGuidingIntegrator
    from practical_path_guiding_lab_amd.render import IndependentSampler, WavefrontScene
    from practical_path_guiding_lab_amd.scene import cornell_box

    w, h, seed = 37, 23, 977 + spp
    sc = cornell_box(w, h, 3, 8)
    g = PathGuidingIntegrator({"max_depth": 3})
    g.setup(w * h, sc.bbox_min - 1e-4, sc.bbox_max + 1e-4, 20, 20, True, 0.5)
    g.setIteration(0, True)
    L, _, _ = g.sample(WavefrontScene(sc), IndependentSampler(spp, seed))
    t = g.sdTree
    Lh = L.cpu().numpy()
    for filt, name in ((0, "tent"), (1, "gaussian")):
        img = torch.empty((3, w * h), dtype=torch.float32, device="cuda")
        N.check(t._h, t._lib.pg_film(t._h, filt, seed, spp, L.data_ptr(), img.data_ptr(), None))
        exp = po.film(name, seed, spp, w, h, Lh)
        np.testing.assert_array_equal(img.cpu().numpy().view(np.uint32), exp.view(np.uint32), err_msg=f"{name} spp={spp}")


@pytest.mark.parametrize("which", ["cornell-box", "veach-mis"])
def test_fused_kernel_and_split_pipeline_are_two_implementations_of_one_bounce(which):
    """A quad scene runs the fused k_bounce by default and the split pipeline (the kernels of the mesh
    scenes: ray casting, shading, k_wave_guide, ...) on request (pg_render_split_pipeline).  The two share
    the device functions of the SD-tree and of the BSDFs but nothing of their control flow, state layout
    or record order -- and end with the same radiance per lane, the same sums and the same trees at full
    bench size, without the oracle in the loop."""
    import torch
    from practical_path_guiding_lab_amd import scene as S
    from practical_path_guiding_lab_amd.integrator import PathGuidingIntegrator
    from practical_path_guiding_lab_amd.render import IndependentSampler, WavefrontScene

    sc = S.cornell_box(512, 512, 8, 8) if which == "cornell-box" else S.veach_mis(1280, 720, 3, 8)
    npix = sc.camera.width * sc.camera.height
    bmin, bmax = sc.bbox_min - np.float32(1e-4), sc.bbox_max + np.float32(1e-4)
    runs = []
    for split in (False, True):
        g = PathGuidingIntegrator({"max_depth": sc.max_depth, "rr_depth": sc.rr_depth})
        g.setup(npix, bmin, bmax, 20, 20, True, 0.5)
        ws = WavefrontScene(sc, split_pipeline=split)
        g.sdTree.enableKernelTiming(True)
        cumm, last = 0, None
        for k in range(4):
            g.setIteration(k, False)
            spp = 2 ** (k + 2)
            for s0 in range(0, spp, 4):
                last, _, _ = g.sample(ws, IndependentSampler(4, 77 + cumm))
                cumm += 4
            g.refineAndPrepareSDTreeForNextIteration()
        kt = g.sdTree.readKernelTiming()
        assert (kt.trace_launches > 0) == split   # really the other kernels (k_wave_trace runs in the split pipeline only)
        assert kt.guide_launches == 0             # (k_wave_guide runs on request only: pg_render_stages(2))
        runs.append((last.clone(), g.sumL.clone(), g.sumL2.clone(), g.sdTree.export()))
        del g, ws
        torch.cuda.empty_cache()
    (La, sa, s2a, ta), (Lb, sb, s2b, tb) = runs
    assert torch.equal(La.view(torch.int32), Lb.view(torch.int32))
    assert torch.equal(sa.view(torch.int32), sb.view(torch.int32)) and torch.equal(s2a.view(torch.int32), s2b.view(torch.int32))
    _same_tree(ta, tb)
    assert ta["kdtree_depth"].shape[0] > 100   # a trained tree, not the initial leaf


@pytest.mark.parametrize("which", ["veach-ajar", "cornell-box", "mixed"])
def test_two_passes_in_flight_equal_one_at_a_time(which):
    """Scheduling switches change no result.  WavefrontScene(in_flight=2): consecutive passes alternate between the two buffer sets of pg_pass_params.slot on
    two streams of their own, two on the device at once, with pg_render_overlap on top.  Radiance of every pass, the
    per-pixel sums (added in issue order: fp32), the accumulators and the refined trees equal the
    one-pass-at-a-time run bit for bit over a guided lifecycle of many small passes."""
    import torch
    from practical_path_guiding_lab_amd import scene as S
    from practical_path_guiding_lab_amd.integrator import PathGuidingIntegrator
    from practical_path_guiding_lab_amd.render import IndependentSampler, WavefrontScene

    sc = {"veach-ajar": lambda: S.veach_ajar(96, 54), "cornell-box": lambda: S.cornell_box(64, 64, 8, 8),
          "mixed": lambda: mixed_scene(48)}[which]()
    npix = sc.camera.width * sc.camera.height
    bmin, bmax = sc.bbox_min - np.float32(1e-4), sc.bbox_max + np.float32(1e-4)

    def run(in_flight, overlap, sort=False, stages=0, split=False):
        g = PathGuidingIntegrator({"max_depth": sc.max_depth, "rr_depth": sc.rr_depth})
        g.setup(npix, bmin, bmax, 20, 20, True, 0.5)
        ws = WavefrontScene(sc, in_flight=in_flight, overlap=overlap, sort=sort, stages=stages, split_pipeline=split)
        g.sdTree.enableKernelTiming(True)
        out = []
        cumm = 0
        for k in range(4):
            g.setIteration(k, k == 3)
            g.resetVarianceCounter()
            Ls = []
            for _ in range(7):  # an odd number of passes: the sets alternate across iterations too
                L, valid, _ = g.sample(ws, IndependentSampler(2, 900 + cumm))
                Ls.append((L, valid))
                cumm += 2
            ws.join()
            torch.cuda.synchronize()
            out.append([(L.cpu().numpy(), v.cpu().numpy()) for L, v in Ls])
            out.append((g.sumL.cpu().numpy(), g.sumL2.cpu().numpy()))
            out.append(g.sdTree.exportAccumulators())
            if k < 3:
                g.refineAndPrepareSDTreeForNextIteration()
                out.append(g.sdTree.export())
        kt = g.sdTree.readKernelTiming()
        if kt.trace_launches:  # the split pipeline ran: k_wave_guide exactly when asked for (or implied by the overlap)
            assert (kt.guide_launches > 0) == bool(stages == 2 or overlap)
        return out

    def same(a, b):
        if isinstance(a, dict):
            _same_tree(a, b)
        elif isinstance(a, (list, tuple)):
            assert len(a) == len(b)
            for x, y in zip(a, b):
                same(x, y)
        else:
            a, b = np.asarray(a), np.asarray(b)
            np.testing.assert_array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b)

    ref = run(1, 0)
    assert np.isfinite(ref[1][0]).all() and ref[1][0].max() > 0
    same(ref, run(2, 0))
    same(ref, run(2, 1))
    same(ref, run(1, 1))
    # pg_render_sort: the bounces below rr_depth in a global spatial order (mesh scenes; a no-op for the fused kernels)
    same(ref, run(1, 0, sort=True))
    same(ref, run(2, 0, sort=True))
    # pg_render_stages: one shading kernel per bounce, three, or four with the SD-tree calls as k_wave_guide
    for st in (0, 1, 2):
        same(ref, run(1, 0, sort=True, stages=st, split=True))
        same(ref, run(2, 0, sort=False, stages=st, split=True))


@pytest.mark.parametrize("which", ["cornell-box", "veach-ajar", "torus"])
def test_batched_launch_equals_separate_one_sample_passes(which):
    """pg_pass_params.batched / pg_film_batched: B consecutive one-sample passes (the reference's training passes,
    main.py:192, seeded initial_seed + cumm_spp, :218) traced as one wavefront are those B passes, bit for bit --
    radiance per sample, the developed image of every pass, the per-pixel sums (added in pass order), the accumulators
    of sdTree_current and the tree refined from them -- for the fused kernels (cornell-box, tent film), the wavefront
    pipeline (veach-ajar) and the gaussian film (torus); unguided and guided iterations; a whole film and a band-sharded
    tile."""
    import torch
    from practical_path_guiding_lab_amd import scene as S
    from practical_path_guiding_lab_amd.integrator import PathGuidingIntegrator
    from practical_path_guiding_lab_amd.render import IndependentSampler, WavefrontScene, render, render_batched

    sc = {"cornell-box": lambda: S.cornell_box(40, 28, 6, 3), "veach-ajar": lambda: S.veach_ajar(48, 27, 9),
          "torus": lambda: S.torus(40, 30, 10)}[which]()
    w, h = sc.camera.width, sc.camera.height
    B = 5

    def fresh():
        g = PathGuidingIntegrator({"max_depth": sc.max_depth, "rr_depth": sc.rr_depth})
        g.setup(w * h, sc.bbox_min - np.float32(1e-4), sc.bbox_max + np.float32(1e-4), 20, 20, True, 0.5)
        return g, WavefrontScene(sc)

    (ga, wa), (gb, wb) = fresh(), fresh()
    seed = 1000
    for k in range(4):  # iterations 2 and 3 are guided
        for g in (ga, gb):
            g.setIteration(k, False)
            g.resetVarianceCounter()
        n = 2 ** (k + 2)
        imgs_a, imgs_b = [], []
        for p in range(n):
            imgs_a.append(render(wa, ga, 1, seed + p))
        for p in range(0, n, B):
            imgs_b += list(render_batched(wb, gb, min(B, n - p), seed + p))
        seed += n
        assert len(imgs_a) == len(imgs_b) == n
        for ia, ib in zip(imgs_a, imgs_b):
            assert ia.shape == ib.shape == (h, w, 3)
            assert torch.equal(ia.view(torch.int32), ib.view(torch.int32)), "image of a pass, iteration %d" % k
        assert torch.equal(ga.sumL.view(torch.int32), gb.sumL.view(torch.int32))
        assert torch.equal(ga.sumL2.view(torch.int32), gb.sumL2.view(torch.int32))
        assert torch.equal(ga.sdTree.accumulators(), gb.sdTree.accumulators())
        for g in (ga, gb):
            g.refineAndPrepareSDTreeForNextIteration()
        _same_tree(ga.sdTree.export(), gb.sdTree.export())
    # radiance per sample, and a band-sharded tile (its lanes are the corresponding lanes of the full-frame launch)
    for g in (ga, gb):
        g.setIteration(4, False)
    Ls = [ga.sample(wa, IndependentSampler(1, 77 + s))[0] for s in range(B)]
    Lb = gb.sample(wb, IndependentSampler(B, 77, batched=True))[0].reshape(3, w * h, B)
    for s in range(B):
        assert torch.equal(Ls[s].view(torch.int32), Lb[:, :, s].contiguous().view(torch.int32))
    wb.set_shard(1, 3, 2)
    px = torch.from_numpy(wb.local_pixels()).cuda()
    Lt = gb.sample(wb, IndependentSampler(B, 77, batched=True))[0].reshape(3, -1, B)
    assert torch.equal(Lt.view(torch.int32), Lb[:, px, :].contiguous().view(torch.int32))
    # ... and it is NOT the B-sample pass of the same seed (a different, equally valid set of streams)
    wb.set_shard(0, 1)
    Lm = gb.sample(wb, IndependentSampler(B, 77))[0].reshape(3, w * h, B)
    assert not torch.equal(Lm, Lb)


def test_driver_with_batched_training_launches_equals_the_reference_schedule():
    """run_guided_render(training_spp_per_pass=1), main.py's own schedule, with 1 and with 8 training passes per launch:
    the same image, logs and SD-tree, bit for bit."""
    from practical_path_guiding_lab_amd.driver import run_guided_render
    from practical_path_guiding_lab_amd.integrator import PathGuidingIntegrator
    from practical_path_guiding_lab_amd.render import WavefrontScene
    from practical_path_guiding_lab_amd.scene import cornell_box
    import torch

    def run(launch):
        sc = cornell_box(36, 36, 6, 8)
        g = PathGuidingIntegrator({"max_depth": 6, "rr_depth": 8})
        gt = torch.full((3, 36 * 36), 0.25, device="cuda")
        res = run_guided_render(WavefrontScene(sc), g, budget_spp=124, initial_seed=9, ground_truth=gt, training_spp_per_pass=1,
                                batch_spp=4, log=lambda s: None, training_passes_per_launch=launch)
        rows = {k: np.array(v.rows, dtype=np.float64)[:, 1:] for k, v in res["records"].items() if v.rows}
        return res["image"].cpu().numpy(), g.sdTree.export(), rows

    img1, tree1, rows1 = run(1)
    img8, tree8, rows8 = run(8)
    np.testing.assert_array_equal(img1.view(np.uint32), img8.view(np.uint32))
    _same_tree(tree1, tree8)
    assert rows1.keys() == rows8.keys()
    for k in rows1:
        np.testing.assert_array_equal(rows1[k], rows8[k], err_msg=k)
