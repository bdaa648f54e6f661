"""Host-side mirror of the reference's Mitsuba plugin class
(takkasila/practical_path_guiding_lab src/path_guiding_integrator.py:27-628): same method names,
argument meaning and error behaviour, with the two KDTree objects replaced by one device-resident
SDTree (libpgsd.so).  Mitsuba is not required: `sample()` drives any scene object that implements
the small wavefront protocol of `practical_path_guiding_lab_amd.render` instead of mi.Scene.

Only what the drivers call is mirrored: setup, setIteration, resetVarianceCounter,
sample, computeVariance, computeMSE, refineAndPrepareSDTreeForNextIteration, saveSDTreeToFile,
loadSDTreeFromFile, saveSDTreeOBJ, aov_names, to_string, and the `max_depth` attribute.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from .sdtree import SDTree

epsilon = 0.00001  # path_guiding_integrator.py:14

_LUM = (0.212671, 0.715160, 0.072169)


def luminance(c: torch.Tensor) -> torch.Tensor:
    """mi.luminance for planar (3, N) colours."""
    return c[0] * _LUM[0] + c[1] * _LUM[1] + c[2] * _LUM[2]


class PathGuidingIntegrator:
    def __init__(self, props: Optional[dict] = None, device: int = 0):
        props = props or {}
        # path_guiding_integrator.py:32-41
        self.max_depth = props.get("max_depth", 30)
        if self.max_depth < 0 and self.max_depth != -1:
            raise Exception('"max_depth" must be set to -1 (infinite) or a value >= 0')
        self.rr_depth = props.get("rr_depth", 8)
        if self.rr_depth < 0:
            raise Exception('"rr_depth" must be set to >= 0')
        self.numRays = 0
        self.array_size = 0
        self.isStoreNEERadiance = False
        self.bsdfSamplingFraction = 0.5
        self.iteration = 0
        self.isFinalIter = False
        self.sdTree = SDTree(device)  # sdTree_prev (values) + sdTree_current (accumulators)
        self.device = self.sdTree.device
        self.sumL = None
        self.sumL2 = None
        self.gt_mask = None
        self._bbox = None
        # passes in flight on streams of their own (WavefrontScene(in_flight=2)): `_epoch` counts the changes a later pass
        # has to see (the scene makes its streams wait for the current one when it moves), `_join()` makes the current
        # stream wait for the passes before anything here reads what they write
        self._epoch = 0
        self._inflight_scene = None
        # the multi-GPU exchange of an iteration, issued ahead of its refine on a stream of its own (beginAccumulatorExchange)
        self._exchange_stream = None
        self._exchange_pending = False
        self._exchange_done = False

    # ---- path_guiding_integrator.py:77-105 ---------------------------------------------------
    def _join(self) -> None:
        if self._inflight_scene is not None:
            self._inflight_scene.join()
        self._epoch += 1

    def setup(self, numRays: int, bbox_min, bbox_max, sdTreeMaxDepth: int = 10, quadTreeMaxDepth: int = 30,
              isStoreNEERadiance: bool = True, bsdfSamplingFraction: float = 0.5) -> None:
        self._join()
        if int(numRays) != self.numRays:
            self.gt_mask = None  # (one entry per film pixel: a mask does not outlive a change of the film)
        self.numRays = int(numRays)
        self.array_size = self.numRays * self.max_depth
        self.isStoreNEERadiance = bool(isStoreNEERadiance)
        self.bsdfSamplingFraction = float(bsdfSamplingFraction)
        self._bbox = (tuple(float(v) for v in bbox_min), tuple(float(v) for v in bbox_max))
        self.sdTree.setup(bbox_min, bbox_max, self.numRays, self.max_depth, sdTreeMaxDepth, quadTreeMaxDepth,
                          isStoreNEERadiance, bsdfSamplingFraction)
        self.resetVarianceCounter()

    def resetVarianceCounter(self) -> None:  # :108-110
        self._join()
        # (the ground-truth mask is NOT part of the counters: every driver resets them at the top of an iteration)
        self.sumL = torch.zeros((3, max(self.numRays, 1)), dtype=torch.float32, device=self.device)
        self.sumL2 = torch.zeros_like(self.sumL)

    def setIteration(self, iteration: int, isFinalIter: bool) -> None:  # :121-123
        self._join()
        if self._exchange_pending:  # (an exchange whose refine never came: the next iteration must not race with it)
            torch.cuda.current_stream().wait_stream(self._exchange_stream)
            self._exchange_pending = False
        self._exchange_done = False
        self.iteration = int(iteration)
        self.isFinalIter = bool(isFinalIter)
        self.sdTree.setIteration(self.iteration, self.isFinalIter)

    # ---- the render hook (:126-431) ------------------------------------------------------------
    def sample(self, scene, sampler, ray=None, medium=None, active=True, aovs=None):
        """One pass over all pixels.  `scene` must implement the wavefront protocol of
        practical_path_guiding_lab_amd.render (trace_pass); returns (L (3,N), valid (N,), [1])."""
        if not hasattr(scene, "trace_pass"):
            raise TypeError("scene must provide trace_pass(integrator, sampler): see practical_path_guiding_lab_amd.render")
        # the pass also adds its samples to sumL / sumL2 on the device (:400-429)
        L, valid, _ = scene.trace_pass(self, sampler, accumulate=True)
        return L, valid, [1]

    def accumulate(self, L: torch.Tensor, spp_per_pass: int = 1) -> None:
        """sumL / sumL2 bookkeeping (:400-429): lanes of one pixel are adjacent when spp_per_pass > 1."""
        if spp_per_pass == 1:
            self.sumL += L
            self.sumL2 += L * L
        else:
            Lr = L.reshape(3, -1, spp_per_pass)
            self.sumL += Lr.sum(dim=2)
            self.sumL2 += (Lr * Lr).sum(dim=2)

    # ---- metrics (:503-550) ---------------------------------------------------------------------
    # `sums`: whole-film (sumL, sumL2) of a tile-sharded render (parallel.all_reduce_sums) instead of
    # this rank's own arrays
    def setGroundTruthMask(self, mask) -> None:
        """Not in the reference: the pixels (bool, film order) a comparison with the ground truth counts.
        veach-ajar here lacks the six teapot shapes (their mesh files are missing from the reference
        mount), so the rectangle they cover in TungstenRender.exr says nothing about the estimator
        (scene.veach_ajar_mask).  None = every pixel, the reference's behaviour."""
        if mask is None:
            self.gt_mask = None
            return
        m = torch.as_tensor(np.asarray(mask, bool).reshape(-1), device=self.device)
        if m.numel() != max(self.numRays, 1):
            raise ValueError("ground-truth mask must have one entry per film pixel")
        self.gt_mask = m

    def _gt_mean(self, per_pixel: torch.Tensor) -> float:
        m = self.gt_mask
        return float((per_pixel if m is None else per_pixel[m]).mean().item())

    def computeMSE(self, spp: float, groundTruth: torch.Tensor, sums=None) -> float:
        self._join()
        sumL = self.sumL if sums is None else sums[0]
        L = sumL / spp
        mse = (L - groundTruth) ** 2
        mse = torch.clamp(luminance(mse), max=10000.0)
        return self._gt_mean(mse)

    def computeVariance(self, spp: float, groundTruth: Optional[torch.Tensor] = None, sums=None) -> float:
        self._join()
        sumL, sumL2 = (self.sumL, self.sumL2) if sums is None else sums
        if groundTruth is not None:
            variance = (sumL2 / spp) - (groundTruth * groundTruth)
            variance = torch.clamp(luminance(variance), max=10000.0)
            return self._gt_mean(variance) / spp
        L = sumL / spp
        L2 = sumL2 / spp
        variance = torch.clamp(luminance(L2 - L * L), max=10000.0)
        v = float(variance.mean().item())
        if spp > 1:
            v /= spp - 1
        return v

    # ---- refinement (:553-586) ------------------------------------------------------------------
    def beginAccumulatorExchange(self, all_reduce, overlap: bool = False) -> None:
        """Multi-GPU: issues the iteration's all-reduce of sdTree_current's accumulators NOW -- behind everything the passes
        have queued, on a stream of its own -- so that the 300 MB it moves over xGMI travel while the current stream
        develops the film, sums the images and computes the variance that decides whether the refine happens at all
        (main.py:334-377).  refineAndPrepareSDTreeForNextIteration waits for it instead of exchanging again.  Every rank
        must call it at the same point (the driver does so after the last pass of every iteration that is not final).  An
        exchange whose refine never comes (training stops) costs its time and changes nothing anyone reads.

        overlap = False (the default): the current stream waits for the exchange at once, so that whatever collective the
        caller issues next (the image sum, the film's sums: torch.distributed's communicator) runs BEHIND this one on the
        device.  `all_reduce` may be libpgsd's own RCCL communicator (SDTree.allReduce); two communicators with
        collectives in flight at once and no order between them is a documented NCCL / RCCL deadlock hazard, and the
        overlapped form has run over gloo only (no N > 1 hardware so far: DESIGN 7).  overlap = True is for an `all_reduce`
        on the SAME communicator as the caller's other collectives (the driver's default, torch.distributed on one
        group), where the backend orders them itself, or for a node on which the overlap has been validated
        (bench.py --exchange-overlap 1)."""
        self._join()
        if self._exchange_stream is None:
            self._exchange_stream = torch.cuda.Stream(device=self.device)
        self._exchange_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._exchange_stream):
            self._exchange(all_reduce)
        if overlap:
            self._exchange_pending = True
        else:
            torch.cuda.current_stream().wait_stream(self._exchange_stream)
            self._exchange_done = True  # (refineAndPrepareSDTreeForNextIteration must not exchange a second time)

    def _exchange(self, all_reduce) -> None:
        """The iteration's one exchange on the current stream.  `all_reduce` sums an int64 tensor over the ranks in place; it
        is handed sdTree_current's accumulators in their 24-byte exchange format (SDTree.packAccumulators: a quarter fewer
        bytes than the 32-byte device layout, the same sums) and the result is unpacked into the accumulators.  A callable
        with the attribute `exchanges_itself` (libpgsd's own RCCL communicator: SDTree.allReduce packs, sums and unpacks
        inside the library) is called with None."""
        if getattr(all_reduce, "exchanges_itself", False):
            all_reduce(None)
            return
        all_reduce(self.sdTree.packAccumulators())
        self.sdTree.unpackAccumulators()

    def refineAndPrepareSDTreeForNextIteration(self, all_reduce=None) -> None:
        """all_reduce: optional callable(int64 tensor) -> None summing the accumulators over ranks
        (torch.distributed.all_reduce on the RCCL group) before the deterministic refine."""
        self._join()
        if self._exchange_pending:  # (already on its way: beginAccumulatorExchange)
            torch.cuda.current_stream().wait_stream(self._exchange_stream)
            self._exchange_pending = False
        elif self._exchange_done:  # (issued and already ordered ahead of the current stream: overlap = False)
            self._exchange_done = False
        elif all_reduce is not None:
            self._exchange(all_reduce)
        self.sdTree.refineAndPrepare()

    # ---- files (:589-615) -----------------------------------------------------------------------
    def saveSDTreeToFile(self, fileName: str) -> None:
        self._join()
        self.sdTree.saveToFile(fileName)

    def loadSDTreeFromFile(self, fileName: str) -> None:
        self._join()
        self.sdTree.loadFromFile(fileName)
        self.isStoreNEERadiance = self.sdTree.store_nee

    def saveSDTreeOBJ(self, fileName: str) -> None:
        write_kd_obj(self.sdTree.export(), fileName)

    def aov_names(self):  # :618-620
        return ["depth.Y"]

    def to_string(self):  # :623-624
        return "path_guiding_integrator"


def write_kd_obj(tree: dict, fileName: str) -> None:
    """KDTree.saveOBJ (kdtree.py:605-663): one wireframe box per KD node."""
    bmin, bmax = tree["kdtree_bbox_min"], tree["kdtree_bbox_max"]
    sceneName = fileName.split("/")[-1].split(".")[0]
    vertCount = 1
    with open(fileName, "w") as f:
        f.write("# OBJ file of KDTree Bounding Boxes\n")
        f.write(f"o {sceneName}\n")
        for i in range(bmin.shape[0]):
            a, b = bmin[i], bmax[i]
            f.write(f"v {a[0]} {a[1]} {a[2]}\n")
            f.write(f"v {b[0]} {a[1]} {a[2]}\n")
            f.write(f"v {b[0]} {a[1]} {b[2]}\n")
            f.write(f"v {a[0]} {a[1]} {b[2]}\n")
            f.write(f"l {vertCount + 0} {vertCount + 1} {vertCount + 2} {vertCount + 3} {vertCount + 0}\n")
            f.write(f"v {a[0]} {b[1]} {a[2]}\n")
            f.write(f"v {b[0]} {b[1]} {a[2]}\n")
            f.write(f"v {b[0]} {b[1]} {b[2]}\n")
            f.write(f"v {a[0]} {b[1]} {b[2]}\n")
            f.write(f"l {vertCount + 4} {vertCount + 5} {vertCount + 6} {vertCount + 7} {vertCount + 4}\n")
            f.write(f"l {vertCount + 0} {vertCount + 4}\n")
            f.write(f"l {vertCount + 1} {vertCount + 5}\n")
            f.write(f"l {vertCount + 2} {vertCount + 6}\n")
            f.write(f"l {vertCount + 3} {vertCount + 7}\n")
            vertCount += 8


def register_with_mitsuba() -> bool:
    """mi.register_integrator('path_guiding_integrator', ...) (path_guiding_integrator.py:628) when
    Mitsuba 3 is importable; the wavefront protocol still has to be provided by the scene."""
    try:
        import mitsuba as mi  # type: ignore
    except ImportError:
        return False
    mi.register_integrator("path_guiding_integrator",
                           lambda props: PathGuidingIntegrator({k: props[k] for k in props.keys()}))
    return True
