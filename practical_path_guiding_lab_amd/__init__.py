"""MI355X-native SD-tree path guiding: the hot path of takkasila/practical_path_guiding_lab
(KD descent, directional-quadtree sample/pdf, radiance splat, per-iteration refine) as
hand-written HIP kernels for gfx950 behind a C ABI (include/pgsd.h).

Importing the package does not need a GPU; creating an SDTree does (no CPU fallback).
"""
from . import _native  # noqa: F401

__all__ = ["SDTree", "PCG32Sampler", "PathGuidingIntegrator"]


def __getattr__(name):
    if name in ("SDTree", "PCG32Sampler"):
        from . import sdtree
        return getattr(sdtree, name)
    if name == "PathGuidingIntegrator":
        from .integrator import PathGuidingIntegrator
        return PathGuidingIntegrator
    raise AttributeError(name)
