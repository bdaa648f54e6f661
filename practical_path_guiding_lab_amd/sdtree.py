"""Host-side handle of the device-resident SD-tree pair (sdTree_prev / sdTree_current).

Mirrors the method surface the reference integrator uses on its two `KDTree` objects
(takkasila/practical_path_guiding_lab src/kdtree.py; call sites in
src/path_guiding_integrator.py:100-105, 244, 301, 307, 500, 559-563, 575-586, 594, 602-608),
on top of the C ABI of libpgsd.so.  Arrays are torch CUDA tensors in planar layout:
a Vector3f[N] is a float32 tensor of shape (3, N) -- the layout Dr.Jit uses for mi.Vector3f.

PyTorch is used only as the owner of caller-side device buffers and streams.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _native as N


def _stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def _f32(t: torch.Tensor, shape) -> torch.Tensor:
    if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous() or tuple(t.shape) != tuple(shape):
        raise ValueError(f"expected contiguous CUDA float32 tensor of shape {tuple(shape)}, got {t.dtype} {tuple(t.shape)} on {t.device}")
    return t


def _mask(active: Optional[torch.Tensor], n: int):
    if active is None:
        return None, None
    if active.dtype == torch.bool:
        active = active.to(torch.uint8)
    if active.dtype != torch.uint8 or not active.is_cuda or tuple(active.shape) != (n,):
        raise ValueError("mask must be a CUDA bool/uint8 tensor of shape (N,)")
    active = active.contiguous()
    return active, active.data_ptr()


# Debug switch (the parity tests of the stand-alone entry points set it): synchronise the device behind every library call, so
# that a GPU fault is reported under the call that launched the faulting kernel and not under whatever synchronised next
# (round 2's memory-aperture abort surfaced in exportAccumulators, three launches later: profiles/r03/kd_descend_isa/).
SYNC_EVERY_CALL = False


class PCG32Sampler:
    """Per-lane PCG32 streams laid out like Mitsuba's `independent` sampler (state/inc arrays)."""

    def __init__(self, tree: "SDTree", n: int, seed: int = 0, lane0: int = 0):
        self.n = n
        self.state = torch.empty(n, dtype=torch.int64, device=tree.device)
        self.inc = torch.empty(n, dtype=torch.int64, device=tree.device)
        tree._seed(self, seed, lane0)


class SDTree:
    """Both reference trees behind one context: queries read sdTree_prev, recording writes
    sdTree_current; `refineAndPrepare` is path_guiding_integrator.py:566-586."""

    def __init__(self, device: int = 0):
        if not torch.cuda.is_available():
            raise RuntimeError("SDTree needs an MI355X: torch.cuda.is_available() is False and there is no CPU path")
        self._lib = N.lib()
        self.device = torch.device("cuda", device)
        h = C.c_void_p()
        N.check(None, self._lib.pg_create(C.byref(h), device))
        self._h = h
        self.store_nee = True

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.pg_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def _ck(self, rc):
        N.check(self._h, rc)
        if SYNC_EVERY_CALL:  # (a fault of an asynchronous kernel then surfaces HERE, under the call that launched it)
            torch.cuda.synchronize(self.device)

    # ---- lifecycle -----------------------------------------------------------------------
    def setup(self, bbox_min, bbox_max, numRays=0, max_depth=0, sdTreeMaxDepth=10, quadTreeMaxDepth=30,
              isStoreNEERadiance=True, bsdfSamplingFraction=0.5):
        bmin = (C.c_float * 3)(*[float(v) for v in bbox_min])
        bmax = (C.c_float * 3)(*[float(v) for v in bbox_max])
        self.store_nee = bool(isStoreNEERadiance)
        self._ck(self._lib.pg_setup(self._h, bmin, bmax, int(numRays), int(max_depth), int(sdTreeMaxDepth),
                                    int(quadTreeMaxDepth), int(self.store_nee), float(bsdfSamplingFraction)))

    def setIteration(self, iteration: int, isFinalIter: bool = False):
        self._ck(self._lib.pg_set_iteration(self._h, int(iteration), int(bool(isFinalIter))))

    def _seed(self, sampler: PCG32Sampler, seed: int, lane0: int):
        self._ck(self._lib.pg_rng_seed(self._h, sampler.n, seed & 0xFFFFFFFF, lane0 & 0xFFFFFFFF,
                                       sampler.state.data_ptr(), sampler.inc.data_ptr(), _stream_ptr()))

    # ---- queries (kdtree.py:435-496) ------------------------------------------------------
    def getLeafNodeIndex(self, position: torch.Tensor, active: Optional[torch.Tensor] = None) -> torch.Tensor:
        n = position.shape[1]
        _f32(position, (3, n))
        m, mp = _mask(active, n)
        out = torch.empty(n, dtype=torch.int32, device=self.device)
        self._ck(self._lib.pg_get_leaf_node_index(self._h, n, position.data_ptr(), mp, out.data_ptr(), _stream_ptr()))
        return out

    def sample(self, position: torch.Tensor, sampler: PCG32Sampler, active: Optional[torch.Tensor] = None
               ) -> Tuple[torch.Tensor, torch.Tensor]:
        n = position.shape[1]
        _f32(position, (3, n))
        m, mp = _mask(active, n)
        d = torch.empty((3, n), dtype=torch.float32, device=self.device)
        pdf = torch.empty(n, dtype=torch.float32, device=self.device)
        self._ck(self._lib.pg_sample(self._h, n, position.data_ptr(), sampler.state.data_ptr(), sampler.inc.data_ptr(),
                                     mp, d.data_ptr(), pdf.data_ptr(), _stream_ptr()))
        return d, pdf

    def pdf(self, position: torch.Tensor, direction: torch.Tensor, active: Optional[torch.Tensor] = None) -> torch.Tensor:
        n = position.shape[1]
        _f32(position, (3, n))
        _f32(direction, (3, n))
        m, mp = _mask(active, n)
        pdf = torch.empty(n, dtype=torch.float32, device=self.device)
        self._ck(self._lib.pg_pdf(self._h, n, position.data_ptr(), direction.data_ptr(), mp, pdf.data_ptr(), _stream_ptr()))
        return pdf

    def guideBounce(self, position, dir_nee, nee_active, select, dir_io, sampler: PCG32Sampler,
                    pdf_nee_out=None, pdf_out=None, lane_index=None, lane_count=None):
        """The three SD-tree calls of one bounce with one KD descent (pg_guide_bounce).
        lane_index/lane_count: compacted live-ray list from compactLanes (optional)."""
        self.prepareGuideBounce(position, dir_nee, nee_active, select, dir_io, sampler, pdf_nee_out, pdf_out,
                                lane_index, lane_count)()
        return self._last_bounce_out

    def prepareGuideBounce(self, position, dir_nee, nee_active, select, dir_io, sampler: PCG32Sampler,
                           pdf_nee_out=None, pdf_out=None, lane_index=None, lane_count=None):
        """Validates once and returns a zero-argument callable that launches pg_guide_bounce on the
        current stream with the argument list already marshalled (a wavefront loop re-launches the
        same buffers every bounce of every pass; per-call checks would dominate a 30 us kernel)."""
        n = position.shape[1]
        _f32(position, (3, n)); _f32(dir_nee, (3, n)); _f32(dir_io, (3, n))
        na, nap = _mask(nee_active, n)
        sl, slp = _mask(select, n)
        if pdf_nee_out is None:
            pdf_nee_out = torch.empty(n, dtype=torch.float32, device=self.device)
        if pdf_out is None:
            pdf_out = torch.empty(n, dtype=torch.float32, device=self.device)
        lip = lcp = None
        if (lane_index is None) != (lane_count is None):
            raise ValueError("lane_index and lane_count go together")
        if lane_index is not None:
            if (lane_index.dtype != torch.int32 or lane_count.dtype != torch.int32 or lane_index.numel() < n
                    or lane_count.numel() < 2):
                raise ValueError("lane_index must be int32[n], lane_count int32[2] (from compactLanes)")
            lip, lcp = lane_index.data_ptr(), lane_count.data_ptr()
        keep = (position, dir_nee, na, sl, dir_io, sampler, pdf_nee_out, pdf_out, lane_index, lane_count)
        self._last_bounce_out = (pdf_nee_out, pdf_out)
        fn, h = self._lib.pg_guide_bounce, self._h
        args = (h, n, position.data_ptr(), dir_nee.data_ptr(), nap, slp, dir_io.data_ptr(), sampler.state.data_ptr(),
                sampler.inc.data_ptr(), pdf_nee_out.data_ptr(), pdf_out.data_ptr(), lip, lcp)
        ck = self._ck

        def launch(_keep=keep):
            ck(fn(*args, torch.cuda.current_stream().cuda_stream))
        return launch

    def compactLanes(self, select: torch.Tensor, nee_active: Optional[torch.Tensor] = None,
                     idx_out: Optional[torch.Tensor] = None, count_out: Optional[torch.Tensor] = None):
        """Active-ray stream compaction (pg_compact_lanes): returns (idx int32[n], counts int32[2]);
        idx[:counts[0]] are the select==2 lanes, idx[n-counts[1]:] (reversed) the other live lanes."""
        self.prepareCompactLanes(select, nee_active, idx_out, count_out)()
        return self._last_compact_out

    def prepareCompactLanes(self, select: torch.Tensor, nee_active=None, idx_out=None, count_out=None):
        n = select.shape[0]
        s, sp = _mask(select, n)
        ne, nep = _mask(nee_active, n)
        if idx_out is None:
            idx_out = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)
        if count_out is None:
            count_out = torch.zeros(2, dtype=torch.int32, device=self.device)
        if count_out.numel() < 2 or count_out.dtype != torch.int32:
            raise ValueError("count_out must be int32[2]")
        self._last_compact_out = (idx_out, count_out)
        fn, h, ck = self._lib.pg_compact_lanes, self._h, self._ck
        args = (h, n, sp, nep, idx_out.data_ptr(), count_out.data_ptr())

        def launch(_keep=(s, ne, idx_out, count_out)):
            ck(fn(*args, torch.cuda.current_stream().cuda_stream))
        return launch

    # ---- recording (kdtree.py:180-225) ----------------------------------------------------
    def addDataPropagate(self, rec: Dict[str, torch.Tensor], count: Optional[torch.Tensor] = None):
        """rec: position (3,M), direction (2,M), radiance (M,), woPdf (M,), direction_nee (2,M),
        radiance_nee_lum (M,) -- the stream scatterDataIntoSDTree hands to the tree."""
        m = rec["position"].shape[1]
        r = N.pg_records()
        r.position = _f32(rec["position"], (3, m)).data_ptr()
        r.direction = _f32(rec["direction"], (2, m)).data_ptr()
        r.radiance = _f32(rec["radiance"], (m,)).data_ptr()
        r.wo_pdf = _f32(rec["woPdf"], (m,)).data_ptr()
        if self.store_nee:
            r.direction_nee = _f32(rec["direction_nee"], (2, m)).data_ptr()
            r.radiance_nee_lum = _f32(rec["radiance_nee_lum"], (m,)).data_ptr()
        cp = None
        if count is not None:
            if count.dtype != torch.int32 or not count.is_cuda:
                raise ValueError("count must be a CUDA int32 tensor")
            cp = count.data_ptr()
        self._ck(self._lib.pg_splat(self._h, m, C.byref(r), cp, _stream_ptr()))

    def _dense(self, rec: Dict[str, torch.Tensor], S: int) -> N.pg_dense_records:
        d = N.pg_dense_records()
        act = rec["active"]
        if act.dtype == torch.bool:
            act = act.to(torch.uint8)
        rec["_active_u8"] = act.contiguous()
        d.active = rec["_active_u8"].data_ptr()
        d.position = _f32(rec["position"], (3, S)).data_ptr()
        d.direction = _f32(rec["direction"], (2, S)).data_ptr()
        d.bsdf = _f32(rec["bsdf"], (3, S)).data_ptr()
        d.throughput_bsdf = _f32(rec["throughputBsdf"], (3, S)).data_ptr()
        d.throughput_radiance = _f32(rec["throughputRadiance"], (3, S)).data_ptr()
        d.radiance_nee = _f32(rec["radiance_nee"], (3, S)).data_ptr()
        d.direction_nee = _f32(rec["direction_nee"], (2, S)).data_ptr()
        d.wo_pdf = _f32(rec["woPdf"], (S,)).data_ptr()
        return d

    def processRecords(self, num_rays: int, max_depth: int, Lfinal: torch.Tensor, rec: Dict[str, torch.Tensor]):
        """processPathData + the filter of scatterDataIntoSDTree (path_guiding_integrator.py:434-497)."""
        S = num_rays * max_depth
        _f32(Lfinal, (3, num_rays))
        d = self._dense(rec, S)
        out = {
            "position": torch.empty((3, S), dtype=torch.float32, device=self.device),
            "direction": torch.empty((2, S), dtype=torch.float32, device=self.device),
            "radiance": torch.empty(S, dtype=torch.float32, device=self.device),
            "woPdf": torch.empty(S, dtype=torch.float32, device=self.device),
            "direction_nee": torch.empty((2, S), dtype=torch.float32, device=self.device),
            "radiance_nee_lum": torch.empty(S, dtype=torch.float32, device=self.device),
        }
        o = N.pg_records_out()
        o.position = out["position"].data_ptr(); o.direction = out["direction"].data_ptr()
        o.radiance = out["radiance"].data_ptr(); o.wo_pdf = out["woPdf"].data_ptr()
        o.direction_nee = out["direction_nee"].data_ptr(); o.radiance_nee_lum = out["radiance_nee_lum"].data_ptr()
        count = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._ck(self._lib.pg_process_records(self._h, num_rays, max_depth, Lfinal.data_ptr(), C.byref(d), C.byref(o),
                                              count.data_ptr(), _stream_ptr()))
        return out, count

    def processAndSplat(self, num_rays: int, max_depth: int, Lfinal: torch.Tensor, rec: Dict[str, torch.Tensor]):
        self.prepareProcessAndSplat(num_rays, max_depth, Lfinal, rec)()

    def prepareProcessAndSplat(self, num_rays: int, max_depth: int, Lfinal: torch.Tensor, rec: Dict[str, torch.Tensor]):
        S = num_rays * max_depth
        _f32(Lfinal, (3, num_rays))
        d = self._dense(rec, S)
        fn, h, ck = self._lib.pg_process_and_splat, self._h, self._ck
        args = (h, num_rays, max_depth, Lfinal.data_ptr(), C.byref(d))

        def launch(_keep=(Lfinal, rec, d)):
            ck(fn(*args, torch.cuda.current_stream().cuda_stream))
        return launch

    # ---- refinement ---------------------------------------------------------------------
    def refineAndPrepare(self):
        self._ck(self._lib.pg_refine_and_swap(self._h, _stream_ptr()))

    def accumulators(self) -> torch.Tensor:
        """int64 view of sdTree_current's accumulators for torch.distributed.all_reduce (RCCL)."""
        p = C.c_void_p()
        n = C.c_uint64()
        self._ck(self._lib.pg_accumulators(self._h, C.byref(p), C.byref(n)))
        return _wrap_device_i64(p.value, n.value, self.device)

    def sortPlaces(self, keys: torch.Tensor, live: Optional[int] = None) -> torch.Tensor:
        """pg_sort_places: the ordering step of a sorted bounce by itself -- keys: uint16-valued places' keys (an int16 / uint16
        tensor of n entries on this device); returns the uint32 places (as int32 bits) in key order for the first `live` places."""
        n = int(keys.numel())
        k = keys.contiguous()
        if k.element_size() != 2:
            raise TypeError("keys: a 16-bit tensor")
        out = torch.full((n,), -1, dtype=torch.int32, device=self.device)
        d_live = None
        if live is not None:
            d_live = torch.tensor([int(live)], dtype=torch.int32, device=self.device)
        self._ck(self._lib.pg_sort_places(self._h, n, k.data_ptr(), None if d_live is None else d_live.data_ptr(), out.data_ptr(), _stream_ptr()))
        return out

    def packAccumulators(self) -> torch.Tensor:
        """sdTree_current's accumulators in the 24-byte exchange format (pg_exchange_pack, on the current stream): the int64
        tensor a host-side collective sums instead of accumulators() -- a quarter fewer bytes; unpackAccumulators() writes
        the sums back."""
        p = C.c_void_p()
        n = C.c_uint64()
        self._ck(self._lib.pg_exchange_pack(self._h, C.byref(p), C.byref(n), _stream_ptr()))
        return _wrap_device_i64(p.value, n.value, self.device)

    def unpackAccumulators(self) -> None:
        self._ck(self._lib.pg_exchange_unpack(self._h, _stream_ptr()))

    def commInfo(self):
        """(ranks, rank) as RCCL reports them for this tree's communicator (ncclCommCount, ncclCommUserRank)."""
        n, r = C.c_int32(-1), C.c_int32(-1)
        self._ck(self._lib.pg_comm_info(self._h, C.byref(n), C.byref(r)))
        return int(n.value), int(r.value)

    # ---- the library's own exchange (pg_comm_*, pg_allreduce: RCCL bound at run time) ----------
    def commUniqueId(self) -> bytes:
        buf = (C.c_uint8 * 128)()
        self._ck(self._lib.pg_comm_unique_id(self._h, buf))
        return bytes(buf)

    def commInit(self, n_ranks: int, rank: int, unique_id: bytes) -> None:
        """ncclCommInitRank on this tree's device: collective over the n_ranks callers."""
        if len(unique_id) != 128:
            raise ValueError("unique_id: 128 bytes from commUniqueId() of rank 0")
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self._ck(self._lib.pg_comm_init(self._h, int(n_ranks), int(rank), buf))

    def commDestroy(self) -> None:
        self._ck(self._lib.pg_comm_destroy(self._h))

    def allReduce(self) -> None:
        """In-place sum of sdTree_current's accumulators over the ranks (pg_allreduce), asynchronous on
        the current stream; call before refineAndPrepare on every rank."""
        self._ck(self._lib.pg_allreduce(self._h, _stream_ptr()))

    # ---- import / export in the reference's npz schema (kdtree.py:539-602) -----------------
    def sizes(self) -> N.pg_tree_sizes:
        s = N.pg_tree_sizes()
        self._ck(self._lib.pg_export_sizes(self._h, C.byref(s)))
        return s

    def export(self) -> Dict[str, np.ndarray]:
        s = self.sizes()
        nk, nq, nr = s.n_kd, s.n_quad, s.n_roots
        a = {
            "kdtree_bbox_min": np.empty((nk, 3), np.float32), "kdtree_bbox_max": np.empty((nk, 3), np.float32),
            "kdtree_depth": np.empty(nk, np.uint32), "kdtree_vertCount": np.empty(nk, np.float32),
            "kdtree_isLeaf": np.empty(nk, np.uint8), "kdtree_quadTreeRootIndex": np.empty(nk, np.uint32),
            "kdtree_child_left_index": np.empty(nk, np.uint32), "kdtree_child_right_index": np.empty(nk, np.uint32),
            "quadtree_rootNodeIndex": np.empty(nr, np.uint32),
            "quadtree_bbox_min": np.empty((nq, 2), np.float32), "quadtree_bbox_max": np.empty((nq, 2), np.float32),
            "quadtree_depth": np.empty(nq, np.uint32), "quadtree_irradiance": np.empty(nq, np.float32),
            "quadtree_isLeaf": np.empty(nq, np.uint8), "quadtree_refinementThreshold": np.empty(nq, np.float32),
        }
        for k in (1, 2, 3, 4):
            a["quadtree_child_%d_index" % k] = np.empty(nq, np.uint32)
        c = _columns(a)
        self._ck(self._lib.pg_export(self._h, C.byref(s), C.byref(c)))
        a["kdtree_isLeaf"] = a["kdtree_isLeaf"].astype(bool)
        a["quadtree_isLeaf"] = a["quadtree_isLeaf"].astype(bool)
        a["kdtree_maxLeafSize"] = np.float64(c.kd_max_leaf_size)
        a["kdtree_maxDepth"] = np.int64(c.kd_max_depth)
        a["quadtree_maxDepth"] = np.int64(c.quad_max_depth)
        a["quadtree_isStoreNEERadiance"] = np.bool_(c.quad_store_nee)
        return a

    def load(self, d: Dict[str, np.ndarray]):
        a = {
            "kdtree_bbox_min": np.ascontiguousarray(d["kdtree_bbox_min"], np.float32),
            "kdtree_bbox_max": np.ascontiguousarray(d["kdtree_bbox_max"], np.float32),
            "kdtree_depth": np.ascontiguousarray(d["kdtree_depth"], np.uint32),
            "kdtree_vertCount": np.ascontiguousarray(d["kdtree_vertCount"], np.float32),
            "kdtree_isLeaf": np.ascontiguousarray(d["kdtree_isLeaf"], np.uint8),
            "kdtree_quadTreeRootIndex": np.ascontiguousarray(d["kdtree_quadTreeRootIndex"], np.uint32),
            "kdtree_child_left_index": np.ascontiguousarray(d["kdtree_child_left_index"], np.uint32),
            "kdtree_child_right_index": np.ascontiguousarray(d["kdtree_child_right_index"], np.uint32),
            "quadtree_rootNodeIndex": np.ascontiguousarray(d["quadtree_rootNodeIndex"], np.uint32),
            "quadtree_bbox_min": np.ascontiguousarray(d["quadtree_bbox_min"], np.float32),
            "quadtree_bbox_max": np.ascontiguousarray(d["quadtree_bbox_max"], np.float32),
            "quadtree_depth": np.ascontiguousarray(d["quadtree_depth"], np.uint32),
            "quadtree_irradiance": np.ascontiguousarray(d["quadtree_irradiance"], np.float32),
            "quadtree_isLeaf": np.ascontiguousarray(d["quadtree_isLeaf"], np.uint8),
            "quadtree_refinementThreshold": np.ascontiguousarray(d["quadtree_refinementThreshold"], np.float32),
        }
        for k in (1, 2, 3, 4):
            a["quadtree_child_%d_index" % k] = np.ascontiguousarray(d["quadtree_child_%d_index" % k], np.uint32)
        c = _columns(a)
        c.kd_max_leaf_size = float(d["kdtree_maxLeafSize"])
        c.kd_max_depth = int(d["kdtree_maxDepth"])
        c.quad_max_depth = int(d["quadtree_maxDepth"])
        c.quad_store_nee = int(bool(d["quadtree_isStoreNEERadiance"]))
        s = N.pg_tree_sizes(a["kdtree_depth"].shape[0], a["quadtree_depth"].shape[0], a["quadtree_rootNodeIndex"].shape[0])
        self._ck(self._lib.pg_import(self._h, C.byref(s), C.byref(c)))
        self.store_nee = bool(c.quad_store_nee)

    def saveToFile(self, fileName: str):  # kdtree.py:539-602
        np.savez_compressed(fileName, **self.export())

    def loadFromFile(self, fileName: str):  # path_guiding_integrator.py:597-608
        self.load(dict(np.load(fileName)))

    def exportAccumulators(self):
        s = self.sizes()
        kd = np.empty(s.n_kd, np.uint64)
        lo = np.empty(s.n_quad, np.uint64)
        hi = np.empty(s.n_quad, np.int64)
        self._ck(self._lib.pg_export_accumulators(self._h, C.byref(s), kd.ctypes.data, lo.ctypes.data, hi.ctypes.data))
        return kd, lo, hi

    def stats(self) -> N.pg_stats:
        st = N.pg_stats()
        self._ck(self._lib.pg_get_stats(self._h, C.byref(st)))
        return st

    def enableKernelTiming(self, on: bool = True):
        self._ck(self._lib.pg_enable_kernel_timing(self._h, int(on)))

    def readKernelTiming(self, reset: bool = True) -> N.pg_kernel_timing:
        kt = N.pg_kernel_timing()
        self._ck(self._lib.pg_read_kernel_timing(self._h, C.byref(kt), int(reset)))
        return kt

    def evalMath(self, which: str, x: torch.Tensor) -> torch.Tensor:
        """The library's deterministic fp32 exp/log/erf/erfinv/sin/cos, element-wise (pg_math_eval)."""
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        out = torch.empty_like(x)
        self._ck(self._lib.pg_math_eval(self._h, ("exp", "log", "erf", "erfinv", "sin", "cos").index(which), x.numel(),
                                        x.data_ptr(), out.data_ptr(), _stream_ptr()))
        return out

    def renderLiveCounts(self, max_depth: int):
        """Paths alive after each bounce of the last rendered pass (pg_render_live_counts)."""
        out = (C.c_uint32 * int(max_depth))()
        self._ck(self._lib.pg_render_live_counts(self._h, out, int(max_depth)))
        return [int(v) for v in out]

    def enableDepthCounters(self, on=True):
        """on: False / True, or 2 -- a probe build's phase stamps alone (pg_read_shade_phases), no counters' atomics."""
        self._ck(self._lib.pg_enable_depth_counters(self._h, int(on)))

    def readShadePhases(self, reset: bool = True):
        """pg_read_shade_phases: (compiled_in, waves, [cycles of the seven phases of k_wave_shade]) -- zeros unless the library
        is the probe build (csrc/Makefile `probe`, libpgsd_phases.so)."""
        out = (C.c_uint64 * 10)()
        self._ck(self._lib.pg_read_shade_phases(self._h, out, int(reset)))
        return bool(out[0]), int(out[1]), [int(out[2 + i]) for i in range(7)]

    def readDepthCounters(self, reset: bool = True) -> N.pg_depth_counters:
        dc = N.pg_depth_counters()
        self._ck(self._lib.pg_read_depth_counters(self._h, C.byref(dc), int(reset)))
        return dc


def _columns(a: Dict[str, np.ndarray]) -> N.pg_tree_columns:
    c = N.pg_tree_columns()
    c.kd_bbox_min = a["kdtree_bbox_min"].ctypes.data
    c.kd_bbox_max = a["kdtree_bbox_max"].ctypes.data
    c.kd_depth = a["kdtree_depth"].ctypes.data
    c.kd_vert_count = a["kdtree_vertCount"].ctypes.data
    c.kd_is_leaf = a["kdtree_isLeaf"].ctypes.data
    c.kd_quad_root_index = a["kdtree_quadTreeRootIndex"].ctypes.data
    c.kd_child_left = a["kdtree_child_left_index"].ctypes.data
    c.kd_child_right = a["kdtree_child_right_index"].ctypes.data
    c.quad_root_node_index = a["quadtree_rootNodeIndex"].ctypes.data
    c.quad_bbox_min = a["quadtree_bbox_min"].ctypes.data
    c.quad_bbox_max = a["quadtree_bbox_max"].ctypes.data
    c.quad_depth = a["quadtree_depth"].ctypes.data
    c.quad_irradiance = a["quadtree_irradiance"].ctypes.data
    c.quad_is_leaf = a["quadtree_isLeaf"].ctypes.data
    c.quad_threshold = a["quadtree_refinementThreshold"].ctypes.data
    for k in range(4):
        c.quad_child[k] = a["quadtree_child_%d_index" % (k + 1)].ctypes.data
    return c


class _DevMem:
    """Minimal __cuda_array_interface__ carrier so torch can alias library-owned device memory."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {
            "shape": (n,), "typestr": "<i8", "data": (ptr, False), "version": 2, "strides": None,
        }


def _wrap_device_i64(ptr: int, n: int, device) -> torch.Tensor:
    if n == 0:
        return torch.empty(0, dtype=torch.int64, device=device)
    return torch.as_tensor(_DevMem(ptr, n), device=device)
