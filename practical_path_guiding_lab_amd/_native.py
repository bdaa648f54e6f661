"""ctypes binding of libpgsd.so (include/pgsd.h).

The library is the product: if it is missing, cannot be loaded, or finds no gfx950 device, this
module raises.  There is deliberately no CPU or PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
# PGSD_LIBRARY: another build of the same library (tools/build_variant.sh), for A/B measurements
LIB_PATH = os.environ.get("PGSD_LIBRARY") or os.path.join(_PKG, "libpgsd.so")
CSRC = os.path.join(_PKG, "csrc")

PG_OK = 0
PG_ACC_LIMBS = 3
PG_FRAC_BITS = 40


class PgError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libpgsd error {code}: {msg}")
        self.code = code


class pg_records(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "position", "direction", "radiance", "wo_pdf", "direction_nee", "radiance_nee_lum")]


class pg_dense_records(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "active", "position", "direction", "bsdf", "throughput_bsdf", "throughput_radiance",
        "radiance_nee", "direction_nee", "wo_pdf")]


class pg_records_out(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "position", "direction", "radiance", "wo_pdf", "direction_nee", "radiance_nee_lum")]


class pg_tree_sizes(C.Structure):
    _fields_ = [("n_kd", C.c_uint64), ("n_quad", C.c_uint64), ("n_roots", C.c_uint64)]


class pg_tree_columns(C.Structure):
    _fields_ = [
        ("kd_max_leaf_size", C.c_double),
        ("kd_max_depth", C.c_int32), ("quad_max_depth", C.c_int32), ("quad_store_nee", C.c_int32),
        ("kd_bbox_min", C.c_void_p), ("kd_bbox_max", C.c_void_p), ("kd_depth", C.c_void_p),
        ("kd_vert_count", C.c_void_p), ("kd_is_leaf", C.c_void_p), ("kd_quad_root_index", C.c_void_p),
        ("kd_child_left", C.c_void_p), ("kd_child_right", C.c_void_p),
        ("quad_root_node_index", C.c_void_p), ("quad_bbox_min", C.c_void_p), ("quad_bbox_max", C.c_void_p),
        ("quad_depth", C.c_void_p), ("quad_irradiance", C.c_void_p), ("quad_is_leaf", C.c_void_p),
        ("quad_threshold", C.c_void_p), ("quad_child", C.c_void_p * 4),
    ]


class pg_stats(C.Structure):
    _fields_ = [
        ("n_kd_nodes", C.c_uint64), ("n_kd_leaves", C.c_uint64), ("n_quad_records", C.c_uint64),
        ("n_quad_nodes", C.c_uint64), ("n_trees", C.c_uint64),
        ("mean_kd_leaf_depth", C.c_double), ("mean_quad_leaf_depth", C.c_double),
        ("max_kd_depth", C.c_uint32), ("max_quad_depth", C.c_uint32),
        ("bytes_kd", C.c_uint64), ("bytes_quad_records", C.c_uint64), ("bytes_accumulators", C.c_uint64),
        ("bytes_jump_tables", C.c_uint64), ("jump_bits", C.c_uint32), ("kd_grid_bits", C.c_uint32),
    ]


class pg_camera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("axis_x", C.c_float * 3), ("axis_y", C.c_float * 3),
                ("axis_z", C.c_float * 3), ("tan_half_fov_x", C.c_float), ("width", C.c_int32), ("height", C.c_int32)]


class pg_scene_desc(C.Structure):
    _fields_ = [("n_quads", C.c_uint64), ("quads", C.c_void_p), ("n_spheres", C.c_uint64), ("spheres", C.c_void_p),
                ("n_materials", C.c_uint64), ("materials", C.c_void_p), ("n_boxes", C.c_uint64), ("boxes", C.c_void_p),
                ("n_tris", C.c_uint64), ("tris", C.c_void_p), ("n_bvh_nodes", C.c_uint64), ("bvh", C.c_void_p),
                ("n_dir_lights", C.c_uint64), ("dir_lights", C.c_void_p), ("bsphere", C.c_float * 4),
                ("tri_normals", C.c_void_p), ("tri_uvs", C.c_void_p), ("n_textures", C.c_uint64), ("textures", C.c_void_p),
                ("n_texels", C.c_uint64), ("texels", C.c_void_p), ("srgb_lut", C.c_void_p)]


class pg_pass_params(C.Structure):
    _fields_ = [("seed", C.c_uint32), ("spp", C.c_int32), ("rr_depth", C.c_int32), ("slot", C.c_int32),
                ("pixel_begin", C.c_uint64), ("pixel_count", C.c_uint64),
                ("stripe_rows", C.c_uint32), ("stripe_index", C.c_uint32), ("stripe_count", C.c_uint32), ("batched", C.c_uint32)]


class pg_kernel_timing(C.Structure):
    _fields_ = [("bounce_ms", C.c_double), ("splat_ms", C.c_double), ("generate_ms", C.c_double),
                ("finish_ms", C.c_double), ("compact_ms", C.c_double), ("bounce_launches", C.c_uint64),
                ("splat_launches", C.c_uint64),
                ("passes", C.c_uint64),
                ("trace_ms", C.c_double), ("shade_ms", C.c_double), ("shadow_ms", C.c_double), ("guide_ms", C.c_double),
                ("tail_ms", C.c_double), ("trace_launches", C.c_uint64), ("guide_launches", C.c_uint64),
                ("shade_a_ms", C.c_double), ("shade_b_ms", C.c_double), ("sort_ms", C.c_double)]


class pg_depth_counters(C.Structure):
    _fields_ = [("kd_levels", C.c_uint64), ("kd_queries", C.c_uint64),
                ("quad_levels", C.c_uint64), ("quad_queries", C.c_uint64), ("layout_bytes", C.c_uint64)]


# every symbol include/pgsd.h declares (checked by tests/test_abi.py)
ABI_VERSION = 6  # PGSD_ABI_VERSION of include/pgsd.h these bindings match

EXPORTS = (
    "pg_create", "pg_destroy", "pg_last_error", "pg_abi_version", "pg_setup", "pg_set_iteration",
    "pg_get_leaf_node_index", "pg_sample", "pg_pdf", "pg_guide_bounce", "pg_compact_lanes", "pg_rng_seed", "pg_splat",
    "pg_process_records", "pg_process_and_splat", "pg_refine_and_swap", "pg_accumulators",
    "pg_export_sizes", "pg_export", "pg_import", "pg_export_accumulators", "pg_get_stats",
    "pg_enable_depth_counters", "pg_read_depth_counters", "pg_scene_set", "pg_render_pass",
    "pg_enable_kernel_timing", "pg_read_kernel_timing", "pg_render_live_counts", "pg_film_tent",
    "pg_math_eval", "pg_scene_set_ex", "pg_film", "pg_film_stripes", "pg_film_batched", "pg_film_batched_accumulate", "pg_render_overlap", "pg_render_sort", "pg_render_stages",
    "pg_comm_unique_id", "pg_comm_init", "pg_comm_attach", "pg_comm_destroy", "pg_allreduce", "pg_render_reserve",
    "pg_render_split_pipeline", "pg_comm_info", "pg_exchange_pack", "pg_exchange_unpack", "pg_exchange_pack_words",
    "pg_exchange_unpack_words", "pg_sort_places", "pg_debug_fail_alloc", "pg_debug_fail_alloc_pending",
    "pg_read_shade_phases",
)


def source_hash() -> str:
    """sha256 (16 hex digits) over the library's sources (csrc/*.hip, csrc/*.hpp, include/pgsd.h): names the code a
    profile was taken of (profiles/pmc_traffic.json) so that bench.py never pairs old counters with new kernels."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.hpp")))
    files.append(os.path.join(os.path.dirname(_PKG), "include", "pgsd.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def build(force: bool = False) -> str:
    """Compile libpgsd.so in-tree with hipcc --offload-arch=gfx950 (works without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j4"]
    if force:
        cmd.append("-B")
    subprocess.run(cmd, check=True)
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). This package has no fallback path.")
    # One HIP runtime per process: PyTorch bundles its own libamdhip64; importing torch first makes
    # libpgsd.so bind to that copy instead of loading /opt/rocm's next to it (two runtimes in one
    # process cannot both open the device: the second one sees no GPU).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    V, U64, I32, U32 = C.c_void_p, C.c_uint64, C.c_int32, C.c_uint32
    L.pg_last_error.restype = C.c_char_p
    L.pg_last_error.argtypes = [V]
    L.pg_abi_version.restype = C.c_int
    if L.pg_abi_version() != ABI_VERSION:  # (a library left over from an older build: its structs have other layouts)
        raise ImportError(f"{LIB_PATH} has ABI version {L.pg_abi_version()}, these bindings are written for {ABI_VERSION}: "
                          "rebuild it with `python -c 'import __graft_entry__ as g; g.build()'`")
    L.pg_create.argtypes = [C.POINTER(V), C.c_int]
    L.pg_destroy.argtypes = [V]
    L.pg_setup.argtypes = [V, V, V, U64, I32, I32, I32, I32, C.c_float]
    L.pg_set_iteration.argtypes = [V, I32, I32]
    L.pg_get_leaf_node_index.argtypes = [V, U64, V, V, V, V]
    L.pg_sample.argtypes = [V, U64, V, V, V, V, V, V, V]
    L.pg_pdf.argtypes = [V, U64, V, V, V, V, V]
    L.pg_guide_bounce.argtypes = [V, U64, V, V, V, V, V, V, V, V, V, V, V, V]
    L.pg_compact_lanes.argtypes = [V, U64, V, V, V, V, V]
    L.pg_rng_seed.argtypes = [V, U64, U32, U32, V, V, V]
    L.pg_splat.argtypes = [V, U64, C.POINTER(pg_records), V, V]
    L.pg_process_records.argtypes = [V, U64, I32, V, C.POINTER(pg_dense_records), C.POINTER(pg_records_out), V, V]
    L.pg_process_and_splat.argtypes = [V, U64, I32, V, C.POINTER(pg_dense_records), V]
    L.pg_refine_and_swap.argtypes = [V, V]
    L.pg_debug_fail_alloc.argtypes = [C.c_int64]
    L.pg_read_shade_phases.argtypes = [V, C.POINTER(C.c_uint64), I32]
    L.pg_debug_fail_alloc_pending.argtypes = []
    L.pg_accumulators.argtypes = [V, C.POINTER(V), C.POINTER(U64)]
    L.pg_export_sizes.argtypes = [V, C.POINTER(pg_tree_sizes)]
    L.pg_export.argtypes = [V, C.POINTER(pg_tree_sizes), C.POINTER(pg_tree_columns)]
    L.pg_import.argtypes = [V, C.POINTER(pg_tree_sizes), C.POINTER(pg_tree_columns)]
    L.pg_export_accumulators.argtypes = [V, C.POINTER(pg_tree_sizes), V, V, V]
    L.pg_get_stats.argtypes = [V, C.POINTER(pg_stats)]
    L.pg_enable_depth_counters.argtypes = [V, I32]
    L.pg_read_depth_counters.argtypes = [V, C.POINTER(pg_depth_counters), I32]
    L.pg_scene_set.argtypes = [V, U64, V, C.POINTER(pg_camera)]
    L.pg_render_pass.argtypes = [V, C.POINTER(pg_pass_params), V, V, V, V, V]
    L.pg_enable_kernel_timing.argtypes = [V, I32]
    L.pg_read_kernel_timing.argtypes = [V, C.POINTER(pg_kernel_timing), I32]
    L.pg_render_live_counts.argtypes = [V, C.POINTER(C.c_uint32), I32]
    L.pg_film_tent.argtypes = [V, U32, I32, V, V, V]
    L.pg_film.argtypes = [V, I32, U32, I32, V, V, V]
    L.pg_film_stripes.argtypes = [V, I32, U32, I32, V, V, U32, U32, U32, V]
    L.pg_film_batched.argtypes = [V, I32, U32, I32, V, V, U32, U32, U32, V]
    L.pg_film_batched_accumulate.argtypes = [V, I32, U32, I32, V, V, C.c_float, I32, U32, U32, U32, V]
    L.pg_render_overlap.argtypes = [V, I32]
    L.pg_render_sort.argtypes = [V, I32]
    L.pg_render_stages.argtypes = [V, I32]
    L.pg_math_eval.argtypes = [V, I32, U64, V, V, V]
    L.pg_scene_set_ex.argtypes = [V, C.POINTER(pg_scene_desc), C.POINTER(pg_camera)]
    L.pg_comm_unique_id.argtypes = [V, V]
    L.pg_comm_init.argtypes = [V, I32, I32, V]
    L.pg_comm_attach.argtypes = [V, V, I32]
    L.pg_comm_destroy.argtypes = [V]
    L.pg_allreduce.argtypes = [V, V]
    L.pg_comm_info.argtypes = [V, C.POINTER(I32), C.POINTER(I32)]
    L.pg_exchange_pack.argtypes = [V, C.POINTER(V), C.POINTER(U64), V]
    L.pg_exchange_unpack.argtypes = [V, V]
    L.pg_exchange_pack_words.argtypes = [V, U64, V]
    L.pg_exchange_unpack_words.argtypes = [V, U64, V]
    L.pg_sort_places.argtypes = [V, U64, V, V, V, V]
    L.pg_render_reserve.argtypes = [V, U64]
    L.pg_render_split_pipeline.argtypes = [V, C.c_int32]
    for name in EXPORTS:
        if name not in ("pg_last_error", "pg_abi_version"):
            getattr(L, name).restype = C.c_int
    _lib = L
    return L


def check(ctx, rc: int):
    if rc != PG_OK:
        msg = lib().pg_last_error(ctx)
        raise PgError(rc, msg.decode() if msg else "")
