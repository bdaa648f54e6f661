"""Scene description for the renderer substrate: quads (rectangles and cube faces), spheres, box
primitives and triangle meshes; diffuse, Beckmann rough-conductor / rough-dielectric, smooth
conductor and dielectric BSDFs in a material table (twosided or not); one-sided area emitters on
rectangles and spheres, directional emitters; a perspective camera -- the subset of Mitsuba 3 scene
XML that scenes/cornell-box, scenes/veach-mis and scenes/torus of the reference use.

`load_xml(path)` parses that subset from a Mitsuba 3 XML file (e.g. the reference's own scene
files, when they are available); `cornell_box()`, `veach_mis()` and `torus()` build the same scenes
from their numeric parameters so that tests and the benchmark do not need the files.

This module is plain data preparation (numpy); it is shared by the product and by the tests
that feed the same arrays to the CPU oracle.
"""
from __future__ import annotations

import math
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

QUAD_STRIDE = 24  # floats per quad, layout documented in include/pgsd.h (pg_scene_desc)
SPHERE_STRIDE = 12    # centre 0-2, radius 3, material 4, emitter flag 5, radiance 6-8
MATERIAL_STRIDE = 16  # type 0, reflectance 1-3, alpha 4, eta 5-7, k 8-10, one-sided 11, texture index + 1 (0: none) 12
TEXTURE_STRIDE = 16   # 32-bit words: kind 0 (1 bitmap, 2 checkerboard), width 1, height 2, first texel 3 (u32);
                      # color0 4-6, color1 7-9, to_uv scale 10-11 and offset 12-13 (f32 bit patterns)
TEX_BITMAP, TEX_CHECKERBOARD = 1, 2
BOX_STRIDE = 32       # rows of the inverse linear map 0-8, centre 9-11, +x/+y/+z face normals 12-20, material 21
MAT_DIFFUSE, MAT_ROUGHCONDUCTOR, MAT_CONDUCTOR, MAT_DIELECTRIC, MAT_ROUGHDIELECTRIC = 0, 1, 2, 3, 4
# RGB indices of refraction Mitsuba's `material` presets resolve to in an RGB variant
CONDUCTOR_PRESETS = {"Al": ((1.657460, 0.880369, 0.521229), (9.223869, 6.269523, 4.837001))}
IOR_PRESETS = {"vacuum": 1.0, "air": 1.000277, "water": 1.3330, "acrylic glass": 1.49, "bk7": 1.5046, "diamond": 2.419}


@dataclass
class Camera:
    origin: np.ndarray
    axis_x: np.ndarray
    axis_y: np.ndarray
    axis_z: np.ndarray
    tan_half_fov_x: np.float32
    width: int
    height: int


@dataclass
class Scene:
    quads: np.ndarray                 # (Q, 24) float32
    camera: Camera
    max_depth: int = 30
    rr_depth: int = 8
    bbox_min: np.ndarray = field(default_factory=lambda: np.zeros(3, np.float32))
    bbox_max: np.ndarray = field(default_factory=lambda: np.ones(3, np.float32))
    names: List[str] = field(default_factory=list)
    rfilter: str = "tent"             # film reconstruction filter: "tent" (radius 1 pixel, the reference's scenes) or "box"
    spheres: np.ndarray = field(default_factory=lambda: np.zeros((0, SPHERE_STRIDE), np.float32))  # (S, 12)
    materials: Optional[np.ndarray] = None  # (M, 12); None: quad i is diffuse with quads[i, 16:19]
    boxes: np.ndarray = field(default_factory=lambda: np.zeros((0, BOX_STRIDE), np.float32))        # (B, 32)
    tris: np.ndarray = field(default_factory=lambda: np.zeros((0, 16), np.float32))                 # (T, 16), in BVH leaf order
    bvh: np.ndarray = field(default_factory=lambda: np.zeros((0, 32), np.uint32))                   # (M, 32) four-wide nodes, mesh.py
    dir_lights: np.ndarray = field(default_factory=lambda: np.zeros((0, 8), np.float32))            # (K, 8): direction, irradiance
    tri_normals: Optional[np.ndarray] = None  # (T, 9) vertex normals per triangle in `tris` order; None: face normals
    tri_uvs: Optional[np.ndarray] = None      # (T, 6) texture coordinates uv0 uv1 uv2 per triangle in `tris` order
    textures: np.ndarray = field(default_factory=lambda: np.zeros((0, TEXTURE_STRIDE), np.uint32))  # (NT, 16) descriptors
    texels: np.ndarray = field(default_factory=lambda: np.zeros(0, np.uint32))  # RGBA8 sRGB texels of all bitmaps (R in the low byte)
    srgb_lut: np.ndarray = field(default_factory=lambda: srgb_to_linear_lut())  # (256,) 8-bit sRGB -> linear fp32
    skipped: List[str] = field(default_factory=list)  # shapes of the scene file left out (mesh file missing)

    def bounding_sphere(self) -> np.ndarray:
        """Mitsuba's scene.bbox().bounding_sphere(): centre and radius (x, y, z, r), fp32."""
        lo, hi = self.bbox_min.astype(np.float32), self.bbox_max.astype(np.float32)
        c = ((lo + hi) * np.float32(0.5)).astype(np.float32)
        r = np.float32(np.sqrt(np.float32(((hi - c) ** 2).sum())))
        return np.array([c[0], c[1], c[2], r], np.float32)


def _f32(v):
    return np.asarray(v, dtype=np.float32)


def srgb_to_linear_lut() -> np.ndarray:
    """The 256 linear values of 8-bit sRGB (what `mi.Bitmap.convert(..., srgb_gamma=False)` makes of a
    JPG/PNG texel), computed once in double and rounded to fp32: part of the scene data, so that the
    product and the oracle look the same numbers up."""
    c = np.arange(256, dtype=np.float64) / 255.0
    return np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4).astype(np.float32)


def bitmap_texture(image: np.ndarray, to_uv=(1.0, 1.0, 0.0, 0.0)) -> dict:
    """Mitsuba `bitmap` texture (bilinear, repeat) over an 8-bit sRGB image (H, W, 3); row 0 is v = 0
    (Mitsuba's obj loader has already flipped v).  to_uv: (scale u, scale v, offset u, offset v)."""
    img = np.ascontiguousarray(image, np.uint8)
    if img.ndim != 3 or img.shape[2] != 3 or img.shape[0] == 0 or img.shape[1] == 0:
        raise ValueError("bitmap_texture: need an (H, W, 3) uint8 image")
    return {"kind": TEX_BITMAP, "image": img, "to_uv": tuple(float(x) for x in to_uv)}


def checkerboard_texture(color0=(0.4, 0.4, 0.4), color1=(0.2, 0.2, 0.2), to_uv=(1.0, 1.0, 0.0, 0.0)) -> dict:
    """Mitsuba `checkerboard` texture: color0 where frac(u) > .5 and frac(v) > .5 agree, else color1."""
    return {"kind": TEX_CHECKERBOARD, "color0": tuple(color0), "color1": tuple(color1), "to_uv": tuple(float(x) for x in to_uv)}


def pack_textures(textures: List[dict]) -> Tuple[np.ndarray, np.ndarray]:
    """(descriptors (NT, 16) uint32, texels uint32 RGBA8) of a list of texture dicts."""
    desc = np.zeros((len(textures), TEXTURE_STRIDE), np.uint32)
    chunks, first = [], 0
    for i, t in enumerate(textures):
        f = np.zeros(10, np.float32)
        desc[i, 0] = t["kind"]
        if t["kind"] == TEX_BITMAP:
            img = t["image"]
            h, w = img.shape[:2]
            desc[i, 1], desc[i, 2], desc[i, 3] = w, h, first
            rgba = img[:, :, 0].astype(np.uint32) | (img[:, :, 1].astype(np.uint32) << 8) | (img[:, :, 2].astype(np.uint32) << 16)
            chunks.append(rgba.reshape(-1))
            first += w * h
        else:
            f[0:3], f[3:6] = _f32(t["color0"]), _f32(t["color1"])
        f[6:10] = _f32(t["to_uv"])
        desc[i, 4:14] = f.view(np.uint32)
    if first >= 2 ** 31:
        raise ValueError("more than 2^31 texels")
    texels = np.concatenate(chunks).astype(np.uint32) if chunks else np.zeros(0, np.uint32)
    return desc, texels


def texture_eval(sc: "Scene", index: int, u, v) -> np.ndarray:
    """Reference evaluation of texture `index` at (u, v) in numpy, the arithmetic of the kernels
    (fp32, every operation rounded): used by tests."""
    d = sc.textures[index]
    f = d[4:14].view(np.float32)
    u, v = np.float32(u), np.float32(v)
    uu = np.float32(np.float32(f[6] * u) + f[8])
    vv = np.float32(np.float32(f[7] * v) + f[9])
    if int(d[0]) == TEX_CHECKERBOARD:
        fu, fv = np.float32(uu - np.floor(uu)), np.float32(vv - np.floor(vv))
        return f[0:3].copy() if (fu > 0.5) == (fv > 0.5) else f[3:6].copy()
    w, h, first = int(d[1]), int(d[2]), int(d[3])
    x = np.float32(np.float32(uu * np.float32(w)) - np.float32(0.5))
    y = np.float32(np.float32(vv * np.float32(h)) - np.float32(0.5))
    if not abs(x) < 1e9:
        x = np.float32(0)
    if not abs(y) < 1e9:
        y = np.float32(0)
    fx, fy = np.floor(x), np.floor(y)
    wx1, wy1 = np.float32(x - fx), np.float32(y - fy)
    wx0, wy0 = np.float32(np.float32(1) - wx1), np.float32(np.float32(1) - wy1)
    ix0, iy0 = int(fx) % w, int(fy) % h
    ix1, iy1 = (int(fx) + 1) % w, (int(fy) + 1) % h

    def texel(ix, iy):
        t = int(sc.texels[first + iy * w + ix])
        return np.array([sc.srgb_lut[t & 255], sc.srgb_lut[(t >> 8) & 255], sc.srgb_lut[(t >> 16) & 255]], np.float32)

    top = (texel(ix0, iy0) * wx0 + texel(ix1, iy0) * wx1).astype(np.float32)
    bot = (texel(ix0, iy1) * wx0 + texel(ix1, iy1) * wx1).astype(np.float32)
    return (top * wy0 + bot * wy1).astype(np.float32)


def _quad(o, e1, e2, refl, radiance=None) -> np.ndarray:
    """All derived quantities are computed here once, in fp32, and handed to every consumer."""
    o, e1, e2 = _f32(o), _f32(e1), _f32(e2)
    n = np.cross(e1, e2).astype(np.float32)
    ln = np.float32(np.sqrt(np.float32(np.dot(n, n))))
    q = np.zeros(QUAD_STRIDE, np.float32)
    q[0:3], q[3:6], q[6:9] = o, e1, e2
    q[9:12] = (n / ln).astype(np.float32)
    q[12] = np.float32(1.0) / np.float32(np.dot(e1, e1))
    q[13] = np.float32(1.0) / np.float32(np.dot(e2, e2))
    q[14] = ln  # |e1 x e2| = area of the parallelogram
    q[15] = 1.0 if radiance is not None else 0.0
    q[16:19] = _f32(refl)
    if radiance is not None:
        q[19:22] = _f32(radiance)
    return q


def _xf(m: np.ndarray, p, w=1.0) -> np.ndarray:
    v = m @ np.array([p[0], p[1], p[2], w], np.float64)
    return v[:3]


def rectangle(to_world: np.ndarray, refl, radiance=None) -> List[np.ndarray]:
    """Mitsuba `rectangle`: [-1,1]^2 in the xy-plane, normal +z, transformed by to_world."""
    o = _xf(to_world, (-1, -1, 0))
    e1 = _xf(to_world, (2, 0, 0), 0.0)
    e2 = _xf(to_world, (0, 2, 0), 0.0)
    if np.linalg.det(to_world[:3, :3]) < 0:  # keep e1 x e2 on the side of Mitsuba's transformed normal
        o, e1, e2 = o + e1, -e1, e2
    return [_quad(o, e1, e2, refl, radiance)]


def cube(to_world: np.ndarray, refl) -> List[np.ndarray]:
    """Mitsuba `cube`: [-1,1]^3, six outward-facing faces."""
    faces = [  # (origin, e1, e2) with e1 x e2 pointing outwards
        ((-1, -1, 1), (2, 0, 0), (0, 2, 0)),    # +z
        ((1, -1, -1), (-2, 0, 0), (0, 2, 0)),   # -z
        ((1, -1, 1), (0, 0, -2), (0, 2, 0)),    # +x
        ((-1, -1, -1), (0, 0, 2), (0, 2, 0)),   # -x
        ((-1, 1, 1), (2, 0, 0), (0, 0, -2)),    # +y
        ((-1, -1, -1), (2, 0, 0), (0, 0, 2)),   # -y
    ]
    out = []
    mirrored = np.linalg.det(to_world[:3, :3]) < 0
    for o, a, b in faces:
        wo, wa, wb = _xf(to_world, o), _xf(to_world, a, 0.0), _xf(to_world, b, 0.0)
        if mirrored:
            wo, wa = wo + wa, -wa
        out.append(_quad(wo, wa, wb, refl))
    return out


def diffuse_material(reflectance, twosided: bool = True, texture: Optional[int] = None) -> np.ndarray:
    """texture: index into the scene's texture list that replaces `reflectance` on meshes with uvs."""
    m = np.zeros(MATERIAL_STRIDE, np.float32)
    m[0] = MAT_DIFFUSE
    m[1:4] = _f32(reflectance)
    m[11] = 0.0 if twosided else 1.0
    m[12] = 0.0 if texture is None else np.float32(texture + 1)
    return m


def conductor_material(eta, k, specular_reflectance=(1.0, 1.0, 1.0), twosided: bool = False) -> np.ndarray:
    """Mitsuba `conductor`: a perfect mirror weighted by the conductor's Fresnel term (a delta lobe)."""
    m = np.zeros(MATERIAL_STRIDE, np.float32)
    m[0] = MAT_CONDUCTOR
    m[1:4] = _f32(specular_reflectance)
    m[5:8] = _f32(eta)
    m[8:11] = _f32(k)
    m[11] = 0.0 if twosided else 1.0
    return m


def dielectric_material(int_ior: float, ext_ior: float = 1.000277) -> np.ndarray:
    """Mitsuba `dielectric`: smooth interface, delta reflection and refraction, never twosided."""
    m = np.zeros(MATERIAL_STRIDE, np.float32)
    m[0] = MAT_DIELECTRIC
    m[1:4] = 1.0
    m[5] = np.float32(np.float32(int_ior) / np.float32(ext_ior))
    m[11] = 1.0
    return m


def _signed_alpha(alpha, distribution: str) -> np.float32:
    """The material row names the microfacet distribution by the sign of alpha: > 0 beckmann, < 0 ggx."""
    if distribution not in ("beckmann", "ggx"):
        raise ValueError(f"unsupported microfacet distribution {distribution}")
    if not float(alpha) > 0.0:
        raise ValueError("alpha must be > 0")
    return np.float32(alpha) if distribution == "beckmann" else -np.float32(alpha)


def roughdielectric_material(alpha, int_ior: float, ext_ior: float = 1.000277, distribution: str = "beckmann") -> np.ndarray:
    """Mitsuba `roughdielectric`, beckmann or ggx distribution, isotropic alpha, sample_visible: a rough
    interface that reflects and refracts (scenes/torus/scene.xml `glass`); never twosided."""
    m = np.zeros(MATERIAL_STRIDE, np.float32)
    m[0] = MAT_ROUGHDIELECTRIC
    m[1:4] = 1.0
    m[4] = _signed_alpha(alpha, distribution)
    m[5] = np.float32(np.float32(int_ior) / np.float32(ext_ior))
    m[11] = 1.0
    return m


def directional_light(direction, irradiance) -> np.ndarray:
    """Mitsuba `directional` emitter: `direction` is where the light travels (its to_world applied to +z)."""
    d = np.asarray(direction, np.float64)
    out = np.zeros(8, np.float32)
    out[0:3] = (d / np.linalg.norm(d)).astype(np.float32)
    out[3:6] = _f32(irradiance)
    return out


def roughconductor_material(alpha, eta, k, specular_reflectance=(1.0, 1.0, 1.0), distribution: str = "beckmann",
                            texture: Optional[int] = None) -> np.ndarray:
    """Mitsuba `roughconductor`, beckmann or ggx distribution, isotropic alpha, sample_visible (its default).
    texture: index of the texture that replaces `specular_reflectance` on meshes with uvs."""
    m = np.zeros(MATERIAL_STRIDE, np.float32)
    m[0] = MAT_ROUGHCONDUCTOR
    m[1:4] = _f32(specular_reflectance)
    m[4] = _signed_alpha(alpha, distribution)
    m[5:8] = _f32(eta)
    m[8:11] = _f32(k)
    m[12] = 0.0 if texture is None else np.float32(texture + 1)
    return m


def sphere(center, radius, material_index: int, radiance=None) -> np.ndarray:
    s = np.zeros(SPHERE_STRIDE, np.float32)
    s[0:3] = _f32(center)
    s[3] = np.float32(radius)
    s[4] = np.float32(material_index)
    if radiance is not None:
        s[5] = 1.0
        s[6:9] = _f32(radiance)
    return s


def box(to_world: np.ndarray, material_index: int) -> np.ndarray:
    """Mitsuba `cube` ([-1,1]^3 under an affine to_world) as ONE primitive: the renderer intersects
    three slabs in the box's local frame instead of six quads.  Needs a material table."""
    m = np.asarray(to_world, np.float64)
    a = np.linalg.inv(m[:3, :3])
    b = np.zeros(BOX_STRIDE, np.float32)
    b[0:9] = a.reshape(-1).astype(np.float32)
    b[9:12] = m[:3, 3].astype(np.float32)
    for k in range(3):  # outward normal of the +k face: the k-th row of the inverse, normalised
        b[12 + 3 * k:15 + 3 * k] = (a[k] / np.linalg.norm(a[k])).astype(np.float32)
    b[21] = np.float32(material_index)
    b[22:25] = m[:3, 0].astype(np.float32)  # the three half-edges, kept for the bounding box (not read by the kernels)
    b[25:28] = m[:3, 1].astype(np.float32)
    b[28:31] = m[:3, 2].astype(np.float32)
    return b


def _finish(quads: List[np.ndarray], cam: Camera, max_depth: int, rr_depth: int, names, spheres=None, materials=None,
            boxes=None, tris=None, dir_lights=None, textures=None) -> Scene:
    q = np.stack(quads).astype(np.float32) if quads else np.zeros((0, QUAD_STRIDE), np.float32)
    corners = [q[:, 0:3], q[:, 0:3] + q[:, 3:6], q[:, 0:3] + q[:, 6:9], q[:, 0:3] + q[:, 3:6] + q[:, 6:9]]
    s = np.stack(spheres).astype(np.float32) if spheres else np.zeros((0, SPHERE_STRIDE), np.float32)
    if s.shape[0]:
        corners += [s[:, 0:3] - s[:, 3:4], s[:, 0:3] + s[:, 3:4]]
    bx = np.stack(boxes).astype(np.float32) if boxes else np.zeros((0, BOX_STRIDE), np.float32)
    for sx in (-1, 1):
        for sy in (-1, 1):
            for sz in (-1, 1):
                corners.append(bx[:, 9:12] + sx * bx[:, 22:25] + sy * bx[:, 25:28] + sz * bx[:, 28:31])
    # meshes: arrays of triangle records, or (records, vertex normals[, uvs]) tuples from mesh.triangles
    parts = [t if isinstance(t, tuple) else (t,) for t in (tris or [])]
    recs = [t[0] for t in parts]
    smooth = any(len(t) > 1 and t[1] is not None for t in parts)
    mapped = any(len(t) > 2 and t[2] is not None for t in parts)
    tr = np.concatenate(recs).astype(np.float32) if recs else np.zeros((0, 16), np.float32)
    if tr.shape[0]:
        corners += [tr[:, 0:3], tr[:, 0:3] + tr[:, 3:6], tr[:, 0:3] + tr[:, 6:9]]
    corners = np.concatenate(corners)
    sc = Scene(q, cam, max_depth, rr_depth, corners.min(axis=0).astype(np.float32), corners.max(axis=0).astype(np.float32), names)
    if tr.shape[0]:
        from .mesh import build_bvh
        cols = []
        if smooth:  # flat-shaded meshes among smooth ones carry their face normal three times
            cols.append(np.concatenate([t[1] if len(t) > 1 and t[1] is not None else np.tile(t[0][:, 9:12], (1, 3))
                                        for t in parts]).astype(np.float32))
        if mapped:  # meshes without texture coordinates among mapped ones: uv = 0
            cols.append(np.concatenate([t[2] if len(t) > 2 and t[2] is not None else np.zeros((t[0].shape[0], 6), np.float32)
                                        for t in parts]).astype(np.float32))
        if cols:
            sc.bvh, sc.tris, per = build_bvh(tr, np.concatenate(cols, axis=1))
            at = 0
            if smooth:
                sc.tri_normals, at = np.ascontiguousarray(per[:, 0:9]), 9
            if mapped:
                sc.tri_uvs = np.ascontiguousarray(per[:, at:at + 6])
        else:
            sc.bvh, sc.tris = build_bvh(tr)
    sc.spheres = s
    sc.boxes = bx
    if dir_lights:
        sc.dir_lights = np.stack(dir_lights).astype(np.float32)
    sc.materials = np.stack(materials).astype(np.float32) if materials else None
    if textures:
        sc.textures, sc.texels = pack_textures(textures)
    if sc.materials is not None:
        tex = sc.materials[:, 12]
        if (tex < 0).any() or (tex > sc.textures.shape[0]).any() or (tex != np.floor(tex)).any():
            raise ValueError("material texture index out of range")
        used = set(int(v) for v in sc.quads[:, 22]) | set(int(v) for v in s[:, 4]) | set(int(v) for v in bx[:, 21])
        if any(sc.materials[i, 12] != 0 for i in used if 0 <= i < sc.materials.shape[0]):
            raise ValueError("textured materials are built for meshes with texture coordinates only")
        if (tex != 0).any() and tr.shape[0] and sc.tri_uvs is None and (sc.materials[sc.tris[:, 12].astype(int), 12] != 0).any():
            raise ValueError("a textured material is used by a mesh without texture coordinates")
    return sc


def make_camera(to_world: np.ndarray, fov_deg: float, width: int, height: int) -> Camera:
    m = np.asarray(to_world, np.float64)
    return Camera(_f32(m[:3, 3]), _f32(m[:3, 0]), _f32(m[:3, 1]), _f32(m[:3, 2]),
                  np.float32(math.tan(math.radians(fov_deg) / 2.0)), int(width), int(height))


def _mat(values: str) -> np.ndarray:
    return np.array([float(v) for v in values.replace(",", " ").split()], np.float64).reshape(4, 4)


def cornell_box(width: int = 512, height: int = 512, max_depth: int = 8, rr_depth: int = 8, boxes: bool = True) -> Scene:
    """The cornell-box of the reference (scenes/cornell-box/scene.xml: fov 19.5, camera at
    (0,1,6.8) looking down -z, five walls, two boxes, one ceiling light of radiance (17,12,4)).
    boxes=True: the two `cube` shapes are box primitives (and the scene carries a material table);
    boxes=False: six quads each and no material table (the plain pg_scene_set form)."""
    white, red, green = (0.725, 0.71, 0.68), (0.63, 0.065, 0.05), (0.14, 0.45, 0.091)
    shapes = [
        ("Floor", "rectangle", "-4.37114e-008 1 4.37114e-008 0 0 -8.74228e-008 2 0 1 4.37114e-008 1.91069e-015 0 0 0 0 1", white, None),
        ("Ceiling", "rectangle", "-1 7.64274e-015 -1.74846e-007 0 8.74228e-008 8.74228e-008 -2 2 0 -1 -4.37114e-008 0 0 0 0 1", white, None),
        ("BackWall", "rectangle", "1.91069e-015 1 1.31134e-007 0 1 3.82137e-015 -8.74228e-008 1 -4.37114e-008 1.31134e-007 -2 -1 0 0 0 1", white, None),
        ("RightWall", "rectangle", "4.37114e-008 -1.74846e-007 2 1 1 3.82137e-015 -8.74228e-008 1 3.82137e-015 1 2.18557e-007 0 0 0 0 1", green, None),
        ("LeftWall", "rectangle", "-4.37114e-008 8.74228e-008 -2 -1 1 3.82137e-015 -8.74228e-008 1 0 -1 -4.37114e-008 0 0 0 0 1", red, None),
        ("ShortBox", "cube", "0.0851643 0.289542 1.31134e-008 0.328631 3.72265e-009 1.26563e-008 -0.3 0.3 -0.284951 0.0865363 5.73206e-016 0.374592 0 0 0 1", white, None),
        ("TallBox", "cube", "0.286776 0.098229 -2.29282e-015 -0.335439 -4.36233e-009 1.23382e-008 -0.6 0.6 -0.0997984 0.282266 2.62268e-008 -0.291415 0 0 0 1", white, None),
        ("Light", "rectangle", "0.235 -1.66103e-008 -7.80685e-009 -0.005 -2.05444e-008 3.90343e-009 -0.0893 1.98 2.05444e-008 0.19 8.30516e-009 -0.03 0 0 0 1", (0, 0, 0), (17, 12, 4)),
    ]
    quads, names, bxs = [], [], []
    colors = [white, red, green, (0, 0, 0)]
    for name, kind, m, refl, rad in shapes:
        if kind == "cube" and boxes:
            bxs.append(box(_mat(m), colors.index(refl)))
            continue
        qs = rectangle(_mat(m), refl, rad) if kind == "rectangle" else cube(_mat(m), refl)
        for q in qs:
            q[22] = np.float32(colors.index(refl)) if boxes else 0.0
        quads += qs
        names += [name] * len(qs)
    cam = make_camera(_mat("-1 0 0 0 0 1 0 1 0 0 -1 6.8 0 0 0 1"), 19.5, width, height)
    if not boxes:
        return _finish(quads, cam, max_depth, rr_depth, names)
    return _finish(quads, cam, max_depth, rr_depth, names, None, [diffuse_material(c) for c in colors], bxs)


def veach_mis(width: int = 1280, height: int = 720, max_depth: int = 3, rr_depth: int = 8, boxes: bool = True) -> Scene:
    """The veach-mis scene of the reference (scenes/veach-mis/scene.xml: fov 35, four rough-conductor
    plates of alpha 0.01 / 0.05 / 0.1 / 0.25, a diffuse floor and back wall, three sphere lamps of
    radius 1, 0.5 and 0.05 whose radiance grows as their area shrinks), from its numeric parameters."""
    eta, k, spec = (0.200438, 0.924033, 1.10221), (3.91295, 2.45285, 2.14219), (0.3, 0.3, 0.3)
    mats = [diffuse_material((0.5, 0.5, 0.5)), diffuse_material((0.0, 0.0, 0.0))]
    mats += [roughconductor_material(a, eta, k, spec) for a in (0.01, 0.05, 0.1, 0.25)]
    D, N0, SMOOTH, GLOSSY, ROUGH, SUPER = 0, 1, 2, 3, 4, 5
    shapes = [
        ("Smooth", "cube", "0.805757 0.0961775 0 0.264069 -0.673242 0.115108 0 4.09801 0 0 4 0 0 0 0 1", SMOOTH),
        ("Glossy", "cube", "0.972057 0.0567134 0 3.06163 -0.396994 0.138865 0 2.71702 0 0 4 0 0 0 0 1", GLOSSY),
        ("Rough", "cube", "1.03191 0.0277252 0 7.09981 -0.194077 0.147415 0 1.81891 0 0 4 0 0 0 0 1", ROUGH),
        ("Diffuse_0001", "rectangle", "9.9 0 0 4.9 0 -4.32743e-007 9.9 0 0 -23.76 -1.03858e-006 0 0 0 0 1", D),
        ("Diffuse_0002", "rectangle", "-4.32743e-007 -4.32743e-007 9.9 -5 -9.9 1.89158e-014 -4.32743e-007 9.9 0 -23.76 -1.03858e-006 0 0 0 0 1", D),
        ("SuperRough", "cube", "1.04217 0.0182831 0 10.6769 -0.127982 0.148882 0 1.23376 0 0 4 0 0 0 0 1", SUPER),
    ]
    quads, names, bxs = [], [], []
    for name, kind, m, mi in shapes:
        if kind == "cube" and boxes:
            bxs.append(box(_mat(m), mi))
            continue
        qs = rectangle(_mat(m), mats[mi][1:4]) if kind == "rectangle" else cube(_mat(m), mats[mi][1:4])
        for q in qs:
            q[22] = np.float32(mi)
        quads += qs
        names += [name] * len(qs)
    spheres = [sphere((0, 6.5, -2.8), 1.0, N0, (7.59909,) * 3), sphere((0, 6.5, 0), 0.5, N0, (30.3964,) * 3),
               sphere((0, 6.5, 2.7), 0.05, N0, (3039.64,) * 3)]
    cam = make_camera(_mat("-4.37113e-008 0 -1 28.2792 0 1 0 3.5 1 0 -4.37113e-008 1.23612e-006 0 0 0 1"), 35.0, width, height)
    return _finish(quads, cam, max_depth, rr_depth, names, spheres, mats, bxs)


def look_at(origin, target, up) -> np.ndarray:
    """Mitsuba's Transform::look_at: columns left, up', forward, origin."""
    o, t, u = (np.asarray(v, np.float64) for v in (origin, target, up))
    d = (t - o) / np.linalg.norm(t - o)
    left = np.cross(u, d)
    left /= np.linalg.norm(left)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = left, np.cross(d, left), d, o
    return m


def rotation(axis, angle_deg: float) -> np.ndarray:
    """Mitsuba's Transform::rotate (Rodrigues, degrees)."""
    a = np.asarray(axis, np.float64)
    a = a / np.linalg.norm(a)
    s, c = math.sin(math.radians(angle_deg)), math.cos(math.radians(angle_deg))
    k = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    m = np.eye(4)
    m[:3, :3] = c * np.eye(3) + s * k + (1 - c) * np.outer(a, a)
    return m


def _torus_mesh_file() -> str:
    import os
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "torus_meshes.npz")


def torus(width: int = 1024, height: int = 768, max_depth: int = 30, rr_depth: int = 8, meshes: Optional[str] = None) -> Scene:
    """The torus scene of the reference (scenes/torus/scene.xml): a one-sided diffuse donut inside a
    frosted acrylic-glass case (roughdielectric, alpha 0.01) held by two aluminium brackets (smooth
    conductor) on a diffuse floor, lit by one directional light, gaussian film filter, z up.
    `meshes`: the five meshes of meshes.serialized as arrays (data/torus_meshes.npz of this package,
    made by tests/golden/make_torus_fixture.py)."""
    from .mesh import triangles
    data = np.load(meshes or _torus_mesh_file())
    mats = [diffuse_material((0.725, 0.71, 0.68), twosided=False), diffuse_material((0.8, 0.8, 0.4), twosided=False),
            roughdielectric_material(0.01, IOR_PRESETS["acrylic glass"], IOR_PRESETS["air"]),
            conductor_material(*CONDUCTOR_PRESETS["Al"])]
    floor_tw = np.eye(4)
    floor_tw[0, 0], floor_tw[1, 1], floor_tw[0, 3], floor_tw[1, 3] = 0.4, 0.428, 10.0, 24.4
    parts = [("floor", floor_tw, 0), ("donut", np.eye(4), 1), ("glass", np.eye(4), 2), ("metal_a", np.eye(4), 3), ("metal_b", np.eye(4), 3)]
    tris = [triangles(data[f"{n}_v"], data[f"{n}_f"].astype(np.int64), tw, mi, data[f"{n}_n"]) for n, tw, mi in parts]
    sun_tw = rotation((0, 0, 1), -45.0) @ rotation((0, 1, 0), 45.0) @ rotation((0, 1, 0), 180.0)
    sun = directional_light(sun_tw[:3, 2], (2.0, 2.0, 1.8))
    cam = make_camera(look_at((-24.173, -38.184, 30.0076), (-23.7753, -37.4261, 29.4905), (0.261433, 0.446628, 0.855673)),
                      34.6222, width, height)
    sc = _finish([], cam, max_depth, rr_depth, [], None, mats, None, tris, [sun])
    sc.rfilter = "gaussian"
    return sc


def _data_file(name: str) -> str:
    import os
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", name)


# Pixels of the 1280x720 film that show the three teapots of scenes/veach-ajar (Mesh000.obj /
# Mesh009.obj are missing from the reference mount), their shadows and caustics on the table top:
# (x0, y0, x1, y1), half-open; comparisons with the reference's ground truth leave them out.
VEACH_AJAR_TEAPOT_RECT = (270, 370, 850, 590)


def _decode_jpg(raw: np.ndarray) -> np.ndarray:
    """The (H, W, 3) 8-bit sRGB image of a JPG file's bytes: what load_xml's `bitmap` reads from the file itself."""
    import io
    from PIL import Image
    return np.asarray(Image.open(io.BytesIO(np.asarray(raw, np.uint8).tobytes())).convert("RGB"))


def veach_ajar_mask(width: int, height: int) -> np.ndarray:
    """(height, width) bool: True where a veach-ajar image can be compared with TungstenRender.exr."""
    x0, y0, x1, y1 = VEACH_AJAR_TEAPOT_RECT
    m = np.ones((height, width), bool)
    m[int(np.floor(y0 * height / 720.0)):int(np.ceil(y1 * height / 720.0)),
      int(np.floor(x0 * width / 1280.0)):int(np.ceil(x1 * width / 1280.0))] = False
    return m


def veach_ajar(width: int = 1280, height: int = 720, max_depth: int = 13, rr_depth: int = 8, data: Optional[str] = None) -> Scene:
    """The veach-ajar scene of the reference (scenes/veach-ajar/scene.xml): a room lit through a door
    left ajar -- one rectangle emitter of radiance 1000 behind the door, diffuse walls, a textured
    door, table top and landscape picture (`bitmap` textures), a GGX rough-conductor floor whose
    specular reflectance is a `checkerboard`, GGX hinges, a Beckmann door handle (the one smooth-shaded
    mesh), fov 60, tent filter.  Built from its numeric parameters and the package's data file
    (data/veach_ajar.npz, made by tests/golden/make_ajar_fixture.py: the 15 OBJ meshes present in the
    reference mount and the three JPG textures as their files' bytes, decoded here with PIL exactly as
    load_xml decodes the files: full resolution, 1920x1280 / 2000x3008 / 1280x1024).  The six teapot shapes
    (scene.xml:252-293) are left out -- their two mesh files are missing from the reference mount --
    and named in `Scene.skipped`; their three materials stay in the table (unused)."""
    from .mesh import triangles
    d = np.load(data or _data_file("veach_ajar.npz"))
    al_eta, al_k = (1.65746, 0.880369, 0.521229), (9.22387, 6.26952, 4.837)
    # (a data file may also hold decoded (H, W, 3) images as tex_*: round 2's reduced copies, kept readable for A/B runs)
    img = {k: _decode_jpg(d["jpg_" + k]) if ("jpg_" + k) in d.files else d["tex_" + k] for k in ("landscape", "table", "cherry")}
    textures = [bitmap_texture(img["landscape"]), bitmap_texture(img["table"]), bitmap_texture(img["cherry"]),
                checkerboard_texture((0.8, 0.8, 0.8), (0.2, 0.2, 0.2), (20.0, 80.0, 0.0, 0.0))]
    mats = [diffuse_material((0.5, 0.5, 0.5), texture=0),                                     # 0 LandscapeBSDF
            diffuse_material((0.5, 0.5, 0.5), texture=1),                                     # 1 TableBSDF
            roughconductor_material(0.25, al_eta, al_k, (1, 1, 1), "beckmann"),               # 2 DoorHandleBSDF
            diffuse_material((0.5, 0.5, 0.5), texture=2),                                     # 3 DoorBSDF
            diffuse_material((0.8, 0.8, 0.8)),                                                # 4 DiffuseBSDF
            roughconductor_material(0.1, al_eta, al_k, (1, 1, 1), "ggx", texture=3),          # 5 FloorBSDF
            diffuse_material((0.247059, 0.168627, 0.0901961)),                                # 6 DoorFrameBSDF
            diffuse_material((0.258824, 0.207843, 0.145098)),                                 # 7 PictureFrameBSDF
            roughconductor_material(0.1, al_eta, al_k, (1, 1, 1), "ggx"),                     # 8 HingeBSDF
            diffuse_material((0.0, 0.0, 0.0)),                                                # 9 LightBSDF
            roughconductor_material(0.15, al_eta, al_k, (1, 1, 1), "ggx"),                    # 10 Pot2BSDF (teapot, unused)
            dielectric_material(1.5, 1.0),                                                    # 11 MaterialBSDF (teapot, unused)
            diffuse_material((0.8, 0.8, 0.8))]                                                # 12 Pot3BSDF (teapot, unused)
    floor_tw = np.eye(4)
    floor_tw[0, 0], floor_tw[0, 3] = 1.8, 2.3
    eye = np.eye(4)
    shapes = [("Landscape", "Mesh008", eye, 0), ("PictureFrame", "Mesh013", eye, 7), ("Floor", "Mesh011", floor_tw, 5),
              ("DoorHandle", "Mesh015", eye, 2), ("Hinge_0001", "Mesh016", eye, 8), ("Hinge_0002", "Mesh012", eye, 8),
              ("Hinge_0003", "Mesh010", eye, 8), ("Door", "Mesh006", eye, 3), ("DoorFrame", "Mesh005", eye, 6),
              ("Diffuse_0001", "Mesh007", eye, 4), ("Diffuse_0002", "Mesh003", eye, 4), ("Diffuse_0003", "Mesh002", eye, 4),
              ("Diffuse_0004", "Mesh001", eye, 4), ("Table", "Mesh004", eye, 1), ("Diffuse_0005", "Mesh014", floor_tw, 4)]
    tris = []
    for _, mesh, tw, mi in shapes:
        n = d[mesh + "_n"] if (mesh + "_n") in d.files else None  # only the door handle is smooth-shaded
        tris.append(triangles(d[mesh + "_v"], d[mesh + "_f"].astype(np.int64), tw, mi, n, d[mesh + "_uv"]))
    light = rectangle(_mat("0.730445 0 0 -4.4391 0 -1.32136 -1.42138e-007 1.50656 0 1.42138e-007 -1.93037 -4.44377 0 0 0 1"),
                      (0.0, 0.0, 0.0), (1000.0, 1000.0, 1000.0))
    for q in light:
        q[22] = 9.0
    cam = make_camera(_mat("-0.137283 -0.0319925 -0.990015 4.05402 2.71355e-008 0.999478 -0.0322983 1.61647 "
                           "0.990532 -0.00443408 -0.137213 -2.30652 0 0 0 1"), 60.0, width, height)
    sc = _finish(light, cam, max_depth, rr_depth, ["Light"], None, mats, None, tris, None, textures)
    sc.rfilter = "tent"
    sc.skipped = ["Pot2_0001", "Pot2_0002", "Pot3_0001", "Pot3_0002", "Material_0001", "Material_0002"]
    return sc


def load_xml(path: str, width: Optional[int] = None, height: Optional[int] = None, boxes: bool = True,
             skip_missing_meshes: bool = False) -> Scene:
    """Mitsuba 3 XML subset: <default>, perspective sensor (fov, to_world, film size, rfilter); diffuse,
    roughconductor / roughdielectric (beckmann, ggx), conductor and dielectric bsdfs (by id or inline;
    `twosided` wrappers honoured, a bare BSDF is one-sided as in Mitsuba), `bitmap` (bilinear, repeat;
    decoded with PIL) and `checkerboard` textures for a diffuse reflectance or a rough conductor's
    specular reflectance; rectangle / cube shapes, spheres by centre and radius, `serialized` and `obj`
    meshes (with their texture coordinates and, unless face_normals is set, their normals); area
    emitters on rectangles and spheres, `directional` emitters; transforms from matrix / lookat /
    scale / translate / rotate.  Anything else raises ValueError.  boxes: `cube` shapes become box
    primitives (else six quads each).  skip_missing_meshes: a mesh whose file does not exist is left
    out and named in Scene.skipped (scenes/veach-ajar in the reference mount) instead of raising."""
    import os
    root = ET.parse(path).getroot()
    base = os.path.dirname(os.path.abspath(path))
    defaults: Dict[str, str] = {d.get("name"): d.get("value") for d in root.findall("default")}

    def val(s: str) -> str:
        return defaults[s[1:]] if s.startswith("$") else s

    def rgb(node) -> Tuple[float, float, float]:
        v = [float(x) for x in val(node.get("value")).replace(",", " ").split()]
        return tuple(v * 3) if len(v) == 1 else tuple(v)

    def vec(node, default: float) -> List[float]:
        if node.get("value") is not None:
            v = [float(x) for x in val(node.get("value")).replace(",", " ").split()]
            return v * 3 if len(v) == 1 else v
        return [float(val(node.get(a, str(default)))) for a in ("x", "y", "z")]

    def transform(node) -> np.ndarray:
        """to_world: the child operations are applied in document order (each multiplies from the left)."""
        m = np.eye(4)
        for op in (node if node is not None else []):
            if op.tag == "matrix":
                t = _mat(op.get("value"))
            elif op.tag == "lookat":
                t = look_at(*[[float(x) for x in op.get(k).replace(",", " ").split()] for k in ("origin", "target", "up")])
            elif op.tag == "scale":
                t = np.diag(vec(op, 1.0) + [1.0])
            elif op.tag == "translate":
                t = np.eye(4)
                t[:3, 3] = vec(op, 0.0)
            elif op.tag == "rotate":
                t = rotation([float(op.get(a, "0")) for a in ("x", "y", "z")], float(val(op.get("angle"))))
            else:
                raise ValueError(f"unsupported transform element {op.tag}")
            m = t @ m
        return m

    materials: List[np.ndarray] = []
    textures: List[dict] = []

    def texture(node) -> int:
        """Index of the texture for a <texture> element."""
        kind = node.get("type")
        tu = node.find("transform[@name='to_uv']")
        to_uv = (1.0, 1.0, 0.0, 0.0)
        if tu is not None:
            m = transform(tu)
            if abs(m[0, 1]) > 1e-12 or abs(m[1, 0]) > 1e-12:
                raise ValueError("to_uv: only scale and translate are built")
            to_uv = (m[0, 0], m[1, 1], m[0, 3], m[1, 3])  # (a 3-D transform element read as the 2-D one: z is unused)
        if kind == "bitmap":
            ft = node.find("string[@name='filter_type']")
            if ft is not None and ft.get("value") != "bilinear":
                raise ValueError("bitmap: only filter_type bilinear is built")
            wm = node.find("string[@name='wrap_mode']")
            if wm is not None and wm.get("value") != "repeat":
                raise ValueError("bitmap: only wrap_mode repeat is built")
            from PIL import Image
            img = np.asarray(Image.open(os.path.join(base, node.find("string[@name='filename']").get("value"))).convert("RGB"))
            textures.append(bitmap_texture(img, to_uv))
        elif kind == "checkerboard":
            cols = {r.get("name"): rgb(r) for r in node.findall("rgb")}
            textures.append(checkerboard_texture(cols.get("color0", (0.4, 0.4, 0.4)), cols.get("color1", (0.2, 0.2, 0.2)), to_uv))
        else:
            raise ValueError(f"unsupported texture type {kind}")
        return len(textures) - 1

    def ior(node, name: str, default: float) -> float:
        s, f = node.find(f"string[@name='{name}']"), node.find(f"float[@name='{name}']")
        if s is not None:
            if s.get("value") not in IOR_PRESETS:
                raise ValueError(f"unknown index of refraction preset {s.get('value')}")
            return IOR_PRESETS[s.get("value")]
        return float(val(f.get("value"))) if f is not None else default

    def material(node, twosided: bool = False) -> int:
        """Index of the material row for a <bsdf> element."""
        if node.get("type") == "twosided":
            return material(node.find("bsdf"), True)
        kind = node.get("type")
        named = {r.get("name"): rgb(r) for r in node.findall("rgb")}
        tex = {t.get("name"): t for t in node.findall("texture")}
        if any(k not in ("reflectance", "specular_reflectance") for k in tex) or (tex and kind not in ("diffuse", "roughconductor")):
            raise ValueError(f"{kind}: textures are built for a diffuse reflectance and a rough conductor's specular_reflectance")

        def eta_k():
            preset = node.find("string[@name='material']")
            if preset is not None:
                if preset.get("value") not in CONDUCTOR_PRESETS:
                    raise ValueError(f"unknown conductor preset {preset.get('value')}")
                return CONDUCTOR_PRESETS[preset.get("value")]
            if "eta" not in named or "k" not in named:
                raise ValueError(f"{kind}: give eta and k as rgb values or a known `material` preset")
            return named["eta"], named["k"]

        def alpha():
            """(alpha, distribution); Mitsuba's default distribution is beckmann"""
            dist = node.find("string[@name='distribution']")
            a = node.find("float[@name='alpha']")
            return (float(val(a.get("value"))) if a is not None else 0.1), (dist.get("value") if dist is not None else "beckmann")

        if kind == "diffuse":
            materials.append(diffuse_material(named.get("reflectance", (0.5, 0.5, 0.5)), twosided,
                                              texture(tex["reflectance"]) if "reflectance" in tex else None))
        elif kind == "roughconductor":
            if not twosided:
                raise ValueError("roughconductor: only the twosided form is built")
            e, k = eta_k()
            a, dist = alpha()
            materials.append(roughconductor_material(a, e, k, named.get("specular_reflectance", (1.0, 1.0, 1.0)), dist,
                                                     texture(tex["specular_reflectance"]) if "specular_reflectance" in tex else None))
        elif kind == "conductor":
            e, k = eta_k()
            materials.append(conductor_material(e, k, named.get("specular_reflectance", (1.0, 1.0, 1.0)), twosided))
        elif kind == "dielectric":
            materials.append(dielectric_material(ior(node, "int_ior", IOR_PRESETS["bk7"]), ior(node, "ext_ior", IOR_PRESETS["air"])))
        elif kind == "roughdielectric":
            a, dist = alpha()
            materials.append(roughdielectric_material(a, ior(node, "int_ior", IOR_PRESETS["bk7"]), ior(node, "ext_ior", IOR_PRESETS["air"]), dist))
        else:
            raise ValueError(f"unsupported bsdf type {kind}")
        return len(materials) - 1

    bsdfs = {b.get("id"): material(b) for b in root.findall("bsdf")}
    integ = root.find("integrator")
    props = {i.get("name"): int(val(i.get("value"))) for i in integ.findall("integer")} if integ is not None else {}
    sensor = root.find("sensor")
    if sensor is None or sensor.get("type") != "perspective":
        raise ValueError("need a perspective sensor")
    axis = sensor.find("string[@name='fov_axis']")
    if axis is not None and axis.get("value") != "x":
        raise ValueError("only fov_axis = x is built")
    fov = float(val(sensor.find("float[@name='fov']").get("value")))
    film = sensor.find("film")
    fw = int(val(film.find("integer[@name='width']").get("value")))
    fh = int(val(film.find("integer[@name='height']").get("value")))
    cam = make_camera(transform(sensor.find("transform")), fov, width or fw, height or fh)
    quads, names, spheres, bxs, tris, skipped = [], [], [], [], [], []
    for sh in root.findall("shape"):
        kind = sh.get("type")
        ref = sh.find("ref")
        mi = bsdfs[ref.get("id")] if ref is not None else material(sh.find("bsdf"))
        refl = materials[mi][1:4]
        em = sh.find("emitter")
        if em is not None and em.get("type") != "area":
            raise ValueError(f"unsupported emitter type {em.get('type')}")
        rad = rgb(em.find("rgb")) if em is not None else None
        if kind == "sphere":
            if sh.find("transform") is not None:
                raise ValueError("spheres are given by centre and radius (to_world is not built)")
            c, r = sh.find("point[@name='center']"), sh.find("float[@name='radius']")
            center = [float(val(c.get(a, "0"))) for a in ("x", "y", "z")] if c is not None else [0.0, 0.0, 0.0]
            spheres.append(sphere(center, float(val(r.get("value"))) if r is not None else 1.0, mi, rad))
            continue
        m = transform(sh.find("transform"))
        if kind in ("serialized", "obj"):
            if rad is not None:
                raise ValueError("emitting meshes are not supported")
            from .mesh import read_obj, read_serialized, triangles
            fname = os.path.join(base, sh.find("string[@name='filename']").get("value"))
            if skip_missing_meshes and not os.path.exists(fname):
                skipped.append(sh.get("id", fname))
                continue
            uv = None
            if kind == "obj":
                v, f, uv, nrm = read_obj(fname, attributes=True)
            else:
                si = sh.find("integer[@name='shape_index']")
                v, f, nrm = read_serialized(fname, int(val(si.get("value"))) if si is not None else 0)
            fn = sh.find("boolean[@name='face_normals']")
            if fn is not None and fn.get("value") == "true":
                nrm = None
            flip = sh.find("boolean[@name='flip_tex_coords']")
            tris.append(triangles(v, f, m, mi, nrm, uv, flip is None or flip.get("value") == "true"))
            continue
        if kind == "rectangle":
            qs = rectangle(m, refl, rad)
        elif kind == "cube":
            if rad is not None:
                raise ValueError("emitting cubes are not supported")
            if boxes:
                bxs.append(box(m, mi))
                continue
            qs = cube(m, refl)
        else:
            raise ValueError(f"unsupported shape type {kind}")
        for q in qs:
            q[22] = np.float32(mi)
        quads += qs
        names += [sh.get("id", kind)] * len(qs)
    lights = []
    for em in root.findall("emitter"):
        if em.get("type") != "directional":
            raise ValueError(f"unsupported emitter type {em.get('type')}")
        d = em.find("vector[@name='direction']")
        direction = vec(d, 0.0) if d is not None else transform(em.find("transform"))[:3, 2]
        irr = em.find("rgb[@name='irradiance']")
        lights.append(directional_light(direction, rgb(irr) if irr is not None else (1.0, 1.0, 1.0)))
    sc = _finish(quads, cam, props.get("max_depth", 30), props.get("rr_depth", 8), names, spheres, materials, bxs, tris, lights,
                 textures)
    sc.skipped = skipped
    rf = film.find("rfilter")
    sc.rfilter = rf.get("type") if rf is not None else "gaussian"  # hdrfilm's default
    if sc.rfilter not in ("tent", "box", "gaussian"):
        raise ValueError(f"unsupported rfilter type {sc.rfilter}")
    return sc
