"""Renderer-side host glue: the objects a `main.py`-style driver needs around
PathGuidingIntegrator when Mitsuba 3 is not there -- a scene that can trace one pass on the device
(pg_render_pass), a sampler carrying seed and sample count, and `render()` in the role of
`mi.render(scene, spp=..., seed=...)` (main.py:218).

The per-pixel estimate returned by `render()` is the box-filtered mean of the pass's samples; the
reference's metric (computeMSE / computeVariance, path_guiding_integrator.py:503-550) is defined
on the raw per-pixel sums sumL / sumL2, which are accumulated on the device exactly as :400-429.
"""
from __future__ import annotations

import ctypes as C
from typing import Tuple

import numpy as np
import torch

from . import _native as N
from .scene import Scene


class IndependentSampler:
    """What the driver uses of mi.Sampler: `sample_count()` and a seed (PCG32 streams are derived
    per lane on the device: pcg32_seed(seed, lane))."""

    def __init__(self, sample_count: int = 1, seed: int = 0, batched: bool = False):
        """batched: the `sample_count` samples of a pixel are that many consecutive ONE-sample passes with the seeds
        seed, seed + 1, ... traced in one wavefront (pg_pass_params.batched) -- the reference's training passes
        (main.py:192, 218), bit for bit, in one launch instead of sample_count."""
        self._spp = int(sample_count)
        self._seed = int(seed)
        self.batched = bool(batched)

    def sample_count(self) -> int:
        return self._spp

    def set_sample_count(self, spp: int) -> None:
        self._spp = int(spp)

    def seed(self, seed: int) -> None:
        self._seed = int(seed)

    @property
    def seed_value(self) -> int:
        return self._seed


class WavefrontScene:
    """Device-resident scene (quads + camera) implementing the `trace_pass` protocol of
    PathGuidingIntegrator.sample()."""

    def __init__(self, scene: Scene, split_pipeline: bool = False, overlap: int = 0, in_flight: int = 1, sort: bool = True,
                 stages: int = 0):
        """split_pipeline: run the bounce as the split pipeline also for a scene the fused kernel could
        run (pg_render_split_pipeline: same results, the SD-tree queries as a kernel of their own)."""
        self.scene = scene
        self.split_pipeline = bool(split_pipeline)
        self.overlap = int(overlap)  # pg_render_overlap: independent kernels of a pass side by side (same results)
        # pg_render_sort: the live list of a mesh scene's bounce in a global spatial order (same results, +10 % on
        # veach-ajar; the fused kernels of quad scenes ignore it)
        self.sort = bool(sort)
        # pg_render_stages: how a mesh scene's bounce is cut into kernels behind the closest hits -- 0: one (k_wave_shade),
        # 1: k_wave_shade_a with the SD-tree calls | k_wave_cast | k_wave_shade_b, 2: those with k_wave_guide on its own
        # (how the hot path is timed by itself).  Same results.
        if stages not in (0, 1, 2):
            raise ValueError("stages must be 0, 1 or 2")
        self.stages = int(stages)
        # in_flight = 2: consecutive passes alternate between two buffer sets (pg_pass_params.slot) and two streams of
        # their own, so that two are on the device at once (the passes of an iteration are independent, main.py:208-218;
        # same results).  What a pass returns is then valid once join() has made the current stream wait for them.
        if in_flight not in (1, 2):
            raise ValueError("in_flight must be 1 or 2")
        self.in_flight = int(in_flight)
        self._streams = None
        self._n_pass = 0
        self._seen_epoch = None
        self._uploaded_to = None

    # what main.py reads from mi.Scene (main.py:48-53)
    def bbox(self) -> Tuple[np.ndarray, np.ndarray]:
        return self.scene.bbox_min, self.scene.bbox_max

    @property
    def film_size(self) -> Tuple[int, int]:
        return self.scene.camera.width, self.scene.camera.height

    def _upload(self, tree) -> None:
        if self._uploaded_to is tree:
            return
        cam = self.scene.camera
        c = N.pg_camera()
        for k in ("origin", "axis_x", "axis_y", "axis_z"):
            setattr(c, k, (C.c_float * 3)(*[float(v) for v in getattr(cam, k)]))
        c.tan_half_fov_x = float(cam.tan_half_fov_x)
        c.width, c.height = int(cam.width), int(cam.height)
        q = np.ascontiguousarray(self.scene.quads, np.float32)
        s = np.ascontiguousarray(self.scene.spheres, np.float32)
        m = None if self.scene.materials is None else np.ascontiguousarray(self.scene.materials, np.float32)
        b = np.ascontiguousarray(self.scene.boxes, np.float32)
        d = N.pg_scene_desc(q.shape[0], q.ctypes.data if q.size else None, s.shape[0], s.ctypes.data if s.size else None,
                            0 if m is None else m.shape[0], None if m is None else m.ctypes.data,
                            b.shape[0], b.ctypes.data if b.size else None)
        t = np.ascontiguousarray(self.scene.tris, np.float32)
        n = np.ascontiguousarray(self.scene.bvh, np.uint32)
        d.n_tris, d.tris = t.shape[0], (t.ctypes.data if t.size else None)
        d.n_bvh_nodes, d.bvh = n.shape[0], (n.ctypes.data if n.size else None)
        dl = np.ascontiguousarray(self.scene.dir_lights, np.float32)
        d.n_dir_lights, d.dir_lights = dl.shape[0], (dl.ctypes.data if dl.size else None)
        if dl.shape[0]:
            d.bsphere = (C.c_float * 4)(*[float(v) for v in self.scene.bounding_sphere()])
        tn = None if self.scene.tri_normals is None else np.ascontiguousarray(self.scene.tri_normals, np.float32)
        if tn is not None:
            assert tn.shape == (t.shape[0], 9)
            d.tri_normals = tn.ctypes.data
        tu = None if self.scene.tri_uvs is None else np.ascontiguousarray(self.scene.tri_uvs, np.float32)
        tex = np.ascontiguousarray(self.scene.textures, np.uint32)
        txl = np.ascontiguousarray(self.scene.texels, np.uint32)
        lut = np.ascontiguousarray(self.scene.srgb_lut, np.float32)
        if tu is not None:
            assert tu.shape == (t.shape[0], 6)
            d.tri_uvs = tu.ctypes.data
        if tex.shape[0]:
            assert tex.shape[1] == 16 and lut.shape == (256,)
            d.n_textures, d.textures = tex.shape[0], tex.ctypes.data
            d.n_texels, d.texels = txl.shape[0], (txl.ctypes.data if txl.size else None)
            d.srgb_lut = lut.ctypes.data
        N.check(tree._h, tree._lib.pg_render_split_pipeline(tree._h, 1 if self.split_pipeline else 0))
        N.check(tree._h, tree._lib.pg_render_overlap(tree._h, self.overlap))
        N.check(tree._h, tree._lib.pg_render_sort(tree._h, 1 if self.sort else 0))
        N.check(tree._h, tree._lib.pg_render_stages(tree._h, self.stages))
        N.check(tree._h, tree._lib.pg_scene_set_ex(tree._h, C.byref(d), C.byref(c)))
        self._uploaded_to = tree

    # image tile of this rank (multi-GPU): (first pixel, pixel count) in row-major order; None = whole film
    pixel_range = None
    # or: bands of `rows` rows dealt round-robin over `count` ranks, this rank being `index`
    # (pg_pass_params.stripe_*): (rows, index, count); None = pixel_range decides
    stripe = None

    def set_shard(self, rank: int, world: int, stripe_rows: int = 4) -> None:
        """This rank's share of the film for `world` ranks: interleaved bands of stripe_rows rows
        (stripe_rows = 0: one contiguous range of pixels).  world = 1 clears the sharding."""
        self.pixel_range = self.stripe = None
        if world <= 1:
            return
        if stripe_rows > 0:
            self.stripe = (int(stripe_rows), int(rank), int(world))
        else:
            from .parallel import shard
            w, h = self.film_size
            self.pixel_range = shard(w * h, rank, world)

    def local_pixels(self) -> np.ndarray:
        """Film pixel (row-major index) of every tile-local pixel of this rank, in tile order."""
        w, h = self.film_size
        if self.stripe is not None:
            rows, index, count = self.stripe
            r = np.arange(h, dtype=np.int64)
            own = r[(r // rows) % count == index]
            return (own[:, None] * w + np.arange(w, dtype=np.int64)[None, :]).reshape(-1)
        begin, count = self.pixel_range if self.pixel_range is not None else (0, w * h)
        return np.arange(begin, begin + count, dtype=np.int64)

    def reserve(self, integrator, spp: int) -> None:
        """Allocates the pass buffers for passes of up to `spp` samples per pixel of this rank's tile now
        (pg_render_reserve) instead of in the first pass of that size -- as the reference's setup()
        allocates its record arrays (path_guiding_integrator.py:93)."""
        tree = integrator.sdTree
        self._upload(tree)
        n = int(self.local_pixels().shape[0]) * int(spp)
        if n:
            N.check(tree._h, tree._lib.pg_render_reserve(tree._h, n))

    def set_stages(self, integrator, stages: int) -> None:
        """pg_render_stages, from the next pass on (a scheduling switch: results do not change)."""
        self.stages = int(stages)
        tree = integrator.sdTree
        self._upload(tree)
        N.check(tree._h, tree._lib.pg_render_stages(tree._h, self.stages))

    def join(self) -> None:
        """in_flight = 2: the current stream waits for every pass issued so far (no host synchronisation)."""
        if self._streams is not None:
            cur = torch.cuda.current_stream()
            for st in self._streams:
                cur.wait_stream(st)

    @property
    def sharded(self) -> bool:
        return self.stripe is not None or self.pixel_range is not None

    def trace_pass(self, integrator, sampler: IndependentSampler, accumulate: bool = True):
        """One device pass over this rank's tile; returns (L (3,N) float32 cuda, valid (N,) uint8 cuda, spp)."""
        tree = integrator.sdTree
        self._upload(tree)
        cam = self.scene.camera
        spp = sampler.sample_count()
        if self.stripe is not None:
            begin, count = 0, int(self.local_pixels().shape[0])
            stripe = self.stripe
        else:
            begin, count = self.pixel_range if self.pixel_range is not None else (0, cam.width * cam.height)
            stripe = (0, 0, 0)
        n = count * spp
        if accumulate and tuple(integrator.sumL.shape) != (3, cam.width * cam.height):
            # k_finish indexes the sums by film pixel: setup(numRays) must have been given the film size
            raise ValueError(f"integrator.setup(numRays={integrator.sumL.shape[1]}) does not match the film "
                             f"{cam.width}x{cam.height}: call setup again for this scene")
        slot, stream = 0, torch.cuda.current_stream()
        if self.in_flight == 2 and n:
            if self._streams is None:
                self._streams = [torch.cuda.Stream(device=tree.device), torch.cuda.Stream(device=tree.device)]
            slot = self._n_pass & 1
            self._n_pass += 1
            stream = self._streams[slot]
            # whatever the integrator did on the current stream since the last pass (zeroed sums, a refined tree, a new
            # iteration) comes first -- once per such change, not per pass: the passes themselves must not wait for
            # each other through the current stream
            epoch = getattr(integrator, "_epoch", 0)
            if self._seen_epoch != (id(integrator), epoch):
                for st in self._streams:
                    st.wait_stream(torch.cuda.current_stream())
                self._seen_epoch = (id(integrator), epoch)
            integrator._inflight_scene = self
        with torch.cuda.stream(stream):
            L = torch.empty((3, n), dtype=torch.float32, device=tree.device)
            valid = torch.empty(n, dtype=torch.uint8, device=tree.device)
            if n == 0:
                return L, valid, spp
            p = N.pg_pass_params(sampler.seed_value & 0xFFFFFFFF, spp, int(integrator.rr_depth), slot, begin, count,
                                 stripe[0], stripe[1], stripe[2], 1 if getattr(sampler, "batched", False) else 0)
            sl = integrator.sumL.data_ptr() if accumulate else None
            sl2 = integrator.sumL2.data_ptr() if accumulate else None
            N.check(tree._h, tree._lib.pg_render_pass(tree._h, C.byref(p), L.data_ptr(), valid.data_ptr(), sl, sl2,
                                                      stream.cuda_stream))
        if stream is not torch.cuda.current_stream():
            # (the tensors were allocated on the pass's stream and will be read on the current one after join())
            L.record_stream(torch.cuda.current_stream())
            valid.record_stream(torch.cuda.current_stream())
        return L, valid, spp


def render_batched(scene: WavefrontScene, integrator, n_passes: int, seed: int, gather=None, mean_of=None):
    """[render(scene, integrator, 1, seed + s) for s in range(n_passes)] -- the reference's one-sample training passes
    (main.py:192, 218) -- as ONE device pass (pg_pass_params.batched) and one film launch (pg_film_batched): returns the
    n_passes images, (n_passes, H, W, 3), each bit-identical to the image of the separate call; the integrator's sums and
    sdTree_current end as after the separate calls.  Box filter: the image of a one-sample pass is its samples.

    mean_of = (acc, scale): instead of the images, the running mean main.py keeps of an iteration's passes (:218-239) --
    acc (a planar (3, H*W) tensor, or None before the first pass) becomes ((acc + image_0 * scale) + image_1 * scale) + ...,
    fp32, every product and sum rounded on its own, exactly what the host makes of the separate images
    (pg_film_batched_accumulate: the n_passes images are never written); returns acc."""
    sampler = IndependentSampler(n_passes, seed, batched=True)
    L, _, _ = integrator.sample(scene, sampler)
    scene.join()
    w, h = scene.film_size
    tree = integrator.sdTree
    filt = scene.scene.rfilter
    stripes = (0, 0, 0)
    partial = False
    if scene.sharded:
        if gather is None:
            if mean_of is not None:
                raise ValueError("mean_of needs the whole film or a gather")
            return L.reshape(3, -1, n_passes).permute(2, 1, 0).reshape(n_passes, -1, 1, 3).contiguous()
        if hasattr(gather, "reduce_image"):
            partial = True
            stripes = scene.stripe
            if filt in ("tent", "gaussian"):
                L = gather(L, scene, n_passes, 1 if filt == "tent" else 2)
        else:
            L = gather(L, scene, n_passes)
    if mean_of is not None:
        acc, scale = mean_of
        if filt in ("tent", "gaussian"):
            has = acc is not None
            if acc is None:  # (a band-sharded rank develops its own rows only: the others stay zero, as in the separate images)
                acc = (torch.zeros if partial else torch.empty)((3, h * w), dtype=torch.float32, device=tree.device)
            N.check(tree._h, tree._lib.pg_film_batched_accumulate(
                tree._h, ("tent", "gaussian").index(filt), seed & 0xFFFFFFFF, n_passes, L.data_ptr(), acc.data_ptr(),
                C.c_float(float(scale)), 1 if has else 0, stripes[0], stripes[1], stripes[2], torch.cuda.current_stream().cuda_stream))
            return acc
        # box filter: the image of a one-sample pass is its samples
        if partial:
            imgs = torch.zeros((n_passes, 3, h * w), dtype=torch.float32, device=tree.device)
            imgs[:, :, torch.from_numpy(scene.local_pixels()).to(tree.device)] = L.reshape(3, -1, n_passes).permute(2, 0, 1)
        else:
            imgs = L.reshape(3, h * w, n_passes).permute(2, 0, 1)
        for s in range(n_passes):
            wimg = imgs[s] * float(scale)
            acc = wimg if acc is None else acc + wimg
        return acc
    if filt in ("tent", "gaussian"):
        img = (torch.zeros if partial else torch.empty)((n_passes, 3, h * w), dtype=torch.float32, device=tree.device)
        N.check(tree._h, tree._lib.pg_film_batched(tree._h, ("tent", "gaussian").index(filt), seed & 0xFFFFFFFF, n_passes,
                                                   L.data_ptr(), img.data_ptr(), stripes[0], stripes[1], stripes[2],
                                                   torch.cuda.current_stream().cuda_stream))
        return img.reshape(n_passes, 3, h, w).permute(0, 2, 3, 1).contiguous()
    if partial:
        img = torch.zeros((n_passes, 3, h * w), dtype=torch.float32, device=tree.device)
        img[:, :, torch.from_numpy(scene.local_pixels()).to(tree.device)] = L.reshape(3, -1, n_passes).permute(2, 0, 1)
        return img.reshape(n_passes, 3, h, w).permute(0, 2, 3, 1).contiguous()
    return L.reshape(3, h, w, n_passes).permute(3, 1, 2, 0).contiguous()


def render(scene: WavefrontScene, integrator, spp: int, seed: int, gather=None) -> torch.Tensor:
    """mi.render(scene, spp=spp, seed=seed) (main.py:218): one pass, returns the (H, W, 3) image the
    film develops: with the scene's `tent` / `gaussian` reconstruction filter (pg_film) or the
    per-pixel mean (`box`).  The integrator's own sums (computeMSE/computeVariance) are raw per-pixel
    sums either way, as in the reference (:400-429).

    A sharded scene (set_shard) traces this rank's tile only.  gather = parallel.HaloExchange (bands of rows): the
    rank develops its own rows, fetching only the filter's reach beyond them from its ring neighbours, and returns
    a film that is zero outside its rows -- the driver sums those once per iteration (reduce_image).  gather =
    parallel.LaneGather: every rank's lanes are collected into the full-frame lane order first and every rank
    returns the whole image.  Without `gather` a sharded scene returns its tile as (pixels, 1, 3) per-pixel means."""
    sampler = IndependentSampler(spp, seed)
    L, _, _ = integrator.sample(scene, sampler)
    scene.join()  # (in_flight = 2: the film below reads L on the current stream)
    w, h = scene.film_size
    tree = integrator.sdTree
    filt = scene.scene.rfilter
    stripes = (0, 0, 0)
    partial = False
    if scene.sharded:
        if gather is None:
            return L.reshape(3, -1, spp).mean(dim=2).T.reshape(-1, 1, 3).contiguous()
        if hasattr(gather, "reduce_image"):  # parallel.HaloExchange: this rank develops its own rows
            partial = True
            stripes = scene.stripe
            if filt in ("tent", "gaussian"):
                L = gather(L, scene, spp, 1 if filt == "tent" else 2)
        else:
            L = gather(L, scene, spp)
    if filt in ("tent", "gaussian"):
        img = (torch.zeros if partial else torch.empty)((3, h * w), dtype=torch.float32, device=tree.device)
        N.check(tree._h, tree._lib.pg_film_stripes(tree._h, ("tent", "gaussian").index(filt), sampler.seed_value & 0xFFFFFFFF,
                                                   spp, L.data_ptr(), img.data_ptr(), stripes[0], stripes[1], stripes[2],
                                                   torch.cuda.current_stream().cuda_stream))
        return img.reshape(3, h, w).permute(1, 2, 0).contiguous()
    if partial:  # box filter: the per-pixel means of this rank's rows in a film that is zero elsewhere
        img = torch.zeros((3, h * w), dtype=torch.float32, device=tree.device)
        img[:, torch.from_numpy(scene.local_pixels()).to(tree.device)] = L.reshape(3, -1, spp).mean(dim=2)
        return img.reshape(3, h, w).permute(1, 2, 0).contiguous()
    return L.reshape(3, h, w, spp).mean(dim=3).permute(1, 2, 0).contiguous()
