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
MATERIAL_STRIDE = 12  # type 0, reflectance 1-3, alpha 4, eta 5-7, k 8-10
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

    def bounding_sphere(self) -> np.ndarray:
        """Mitsuba's scene.bbox().bounding_sphere(): centre and radius (x, y, z, r), fp32."""
        lo, hi = self.bbox_min.astype(np.float32), self.bbox_max.astype(np.float32)
        c = ((lo + hi) * np.float32(0.5)).astype(np.float32)
        r = np.float32(np.sqrt(np.float32(((hi - c) ** 2).sum())))
        return np.array([c[0], c[1], c[2], r], np.float32)


def _f32(v):
    return np.asarray(v, dtype=np.float32)


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


def diffuse_material(reflectance, twosided: bool = True) -> np.ndarray:
    m = np.zeros(MATERIAL_STRIDE, np.float32)
    m[0] = MAT_DIFFUSE
    m[1:4] = _f32(reflectance)
    m[11] = 0.0 if twosided else 1.0
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


def roughconductor_material(alpha, eta, k, specular_reflectance=(1.0, 1.0, 1.0), distribution: str = "beckmann") -> np.ndarray:
    """Mitsuba `roughconductor`, beckmann or ggx distribution, isotropic alpha, sample_visible (its default)."""
    m = np.zeros(MATERIAL_STRIDE, np.float32)
    m[0] = MAT_ROUGHCONDUCTOR
    m[1:4] = _f32(specular_reflectance)
    m[4] = _signed_alpha(alpha, distribution)
    m[5:8] = _f32(eta)
    m[8:11] = _f32(k)
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
            boxes=None, tris=None, dir_lights=None) -> Scene:
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
    # meshes: arrays of triangle records, or (records, vertex normals) pairs from mesh.triangles
    recs = [t[0] if isinstance(t, tuple) else t for t in (tris or [])]
    smooth = any(isinstance(t, tuple) for t in (tris or []))
    tr = np.concatenate(recs).astype(np.float32) if recs else np.zeros((0, 16), np.float32)
    if tr.shape[0]:
        corners += [tr[:, 0:3], tr[:, 0:3] + tr[:, 3:6], tr[:, 0:3] + tr[:, 6:9]]
    corners = np.concatenate(corners)
    sc = Scene(q, cam, max_depth, rr_depth, corners.min(axis=0).astype(np.float32), corners.max(axis=0).astype(np.float32), names)
    if tr.shape[0]:
        from .mesh import build_bvh
        if smooth:  # flat-shaded meshes among smooth ones carry their face normal three times
            nrm = np.concatenate([t[1] if isinstance(t, tuple) else np.tile(t[:, 9:12], (1, 3)) for t in tris]).astype(np.float32)
            sc.bvh, sc.tris, sc.tri_normals = build_bvh(tr, nrm)
        else:
            sc.bvh, sc.tris = build_bvh(tr)
    sc.spheres = s
    sc.boxes = bx
    if dir_lights:
        sc.dir_lights = np.stack(dir_lights).astype(np.float32)
    sc.materials = np.stack(materials).astype(np.float32) if materials else None
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
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "torus_meshes.npz")


def torus(width: int = 1024, height: int = 768, max_depth: int = 30, rr_depth: int = 8, meshes: Optional[str] = None) -> Scene:
    """The torus scene of the reference (scenes/torus/scene.xml): a one-sided diffuse donut inside a
    frosted acrylic-glass case (roughdielectric, alpha 0.01) held by two aluminium brackets (smooth
    conductor) on a diffuse floor, lit by one directional light, gaussian film filter, z up.
    `meshes`: the five meshes of meshes.serialized as arrays (tests/golden/torus_meshes.npz, made by
    tests/golden/make_torus_fixture.py)."""
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


def load_xml(path: str, width: Optional[int] = None, height: Optional[int] = None, boxes: bool = True) -> Scene:
    """Mitsuba 3 XML subset: <default>, perspective sensor (fov, to_world, film size, rfilter); diffuse,
    roughconductor / roughdielectric (beckmann), conductor and dielectric bsdfs (by id or inline;
    `twosided` wrappers honoured, a bare BSDF is one-sided as in Mitsuba); rectangle / cube shapes,
    spheres by centre and radius, `serialized` and `obj` meshes; area emitters on rectangles and
    spheres, `directional` emitters; transforms from matrix / lookat / scale / translate / rotate.
    Anything else raises ValueError.  boxes: `cube` shapes become box primitives (else six quads
    each)."""
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
            materials.append(diffuse_material(named.get("reflectance", (0.5, 0.5, 0.5)), twosided))
        elif kind == "roughconductor":
            if not twosided:
                raise ValueError("roughconductor: only the twosided form is built")
            e, k = eta_k()
            a, dist = alpha()
            materials.append(roughconductor_material(a, e, k, named.get("specular_reflectance", (1.0, 1.0, 1.0)), dist))
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
    quads, names, spheres, bxs, tris = [], [], [], [], []
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
            if kind == "obj":
                v, f = read_obj(fname)
                nrm = None
            else:
                si = sh.find("integer[@name='shape_index']")
                v, f, nrm = read_serialized(fname, int(val(si.get("value"))) if si is not None else 0)
            fn = sh.find("boolean[@name='face_normals']")
            if fn is not None and fn.get("value") == "true":
                nrm = None
            tris.append(triangles(v, f, m, mi, nrm))
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
    sc = _finish(quads, cam, props.get("max_depth", 30), props.get("rr_depth", 8), names, spheres, materials, bxs, tris, lights)
    rf = film.find("rfilter")
    sc.rfilter = rf.get("type") if rf is not None else "gaussian"  # hdrfilm's default
    if sc.rfilter not in ("tent", "box", "gaussian"):
        raise ValueError(f"unsupported rfilter type {sc.rfilter}")
    return sc
