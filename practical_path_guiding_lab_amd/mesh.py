"""Triangle meshes for the renderer substrate: OBJ reading, triangle records and a binary BVH in
the flat layout the kernels (csrc/pg_render.hip) and the CPU oracle (oracle/pg_oracle_render.c)
traverse -- the `obj` / `serialized` shapes of the reference's scenes (scenes/veach-ajar,
scenes/torus), with face normals.

Layouts (include/pgsd.h):
  triangle, PG_TRI_STRIDE = 16 floats: v0 (0-2), e1 = v1 - v0 (3-5), e2 = v2 - v0 (6-8),
      unit geometric normal (9-11), material index (12), 13-15 unused
  BVH node, PG_BVH_STRIDE = 32 x 32 bit (128 bytes), up to four children:
      0-23  f32: the children's boxes, lo_x[4] lo_y[4] lo_z[4] hi_x[4] hi_y[4] hi_z[4]
      24-27 u32: the children: a node index | 0x80000000 + (count-1) << 28 + first triangle (a leaf
                 of 1..8 triangles) | 0xffffffff (no child)
      28    u32: number of children; 29-31 unused
  Node 0 is the root (a mesh of one leaf still has it), children come after their parent;
  triangles are stored in leaf order.

Plain numpy; shared by the product and by the tests that hand the same arrays to the oracle.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np

TRI_STRIDE = 16
BVH_STRIDE = 32
BVH_HOT_NODES = 96  # numbered first, by box area (see build_bvh)
BVH_WIDTH = 4
LEAF_FLAG = 0x80000000
EMPTY_CHILD = 0xFFFFFFFF
MAX_LEAF = 2  # triangles per leaf at most (measured on veach-ajar: 2 -> 33 ms of ray casting per pass, 3 -> 43, 4 -> 46, 8 -> 55; 1 is too deep for the walk's stack; again with round 3's walks: 2 -> 59.0 ms per step, 3 -> 64.9, 4 -> 66.4)


def read_obj(path: str, attributes: bool = False):
    """Vertices (V,3) float64 and triangles (F,3) int (polygons are fanned).  attributes=True also
    returns the per-corner texture coordinates (F,3,2) and normals (F,3,3) of `v/vt/vn` faces (None
    where the file has none for some face): (v, f, uv, n).  Texture coordinates are returned as the
    file holds them; Mitsuba's `obj` plugin flips v (flip_tex_coords, its default) -- see triangles()."""
    verts: List[List[float]] = []
    tex: List[List[float]] = []
    nrm: List[List[float]] = []
    faces: List[List[int]] = []
    fuv: List[List[int]] = []
    fnr: List[List[int]] = []
    has_uv = has_n = True

    def resolve(tok: str, count: int) -> int:
        i = int(tok)
        return i - 1 if i > 0 else count + i

    with open(path) as f:
        for line in f:
            s = line.split()
            if not s:
                continue
            if s[0] == "v":
                verts.append([float(s[1]), float(s[2]), float(s[3])])
            elif s[0] == "vt":
                tex.append([float(s[1]), float(s[2]) if len(s) > 2 else 0.0])
            elif s[0] == "vn":
                nrm.append([float(s[1]), float(s[2]), float(s[3])])
            elif s[0] == "f":
                parts = [tok.split("/") for tok in s[1:]]
                idx = [resolve(p[0], len(verts)) for p in parts]
                ti = [resolve(p[1], len(tex)) if len(p) > 1 and p[1] else -1 for p in parts]
                ni = [resolve(p[2], len(nrm)) if len(p) > 2 and p[2] else -1 for p in parts]
                for k in range(1, len(idx) - 1):
                    faces.append([idx[0], idx[k], idx[k + 1]])
                    fuv.append([ti[0], ti[k], ti[k + 1]])
                    fnr.append([ni[0], ni[k], ni[k + 1]])
                has_uv = has_uv and min(ti) >= 0
                has_n = has_n and min(ni) >= 0
    v = np.asarray(verts, np.float64).reshape(-1, 3)
    fa = np.asarray(faces, np.int64).reshape(-1, 3)
    if not attributes:
        return v, fa
    uv = np.asarray(tex, np.float64).reshape(-1, 2)[np.asarray(fuv, np.int64).reshape(-1, 3)] if has_uv and tex and faces else None
    n = np.asarray(nrm, np.float64).reshape(-1, 3)[np.asarray(fnr, np.int64).reshape(-1, 3)] if has_n and nrm and faces else None
    return v, fa, uv, n


def read_serialized(path: str, shape_index: int):
    """One mesh of a Mitsuba `.serialized` file (the `serialized` shapes of scenes/torus/scene.xml):
    returns (vertices (V,3) float64, faces (F,3) int64, vertex normals (V,3) float64 or None).
    Layout: per mesh a uint16 magic 0x041C, a uint16 version (3 or 4), then a zlib stream holding
    uint32 flags, (v4) a NUL-terminated name, uint64 vertex and triangle counts, positions, optional
    normals / texcoords / colours (single or double precision by flag), uint32 indices; the file
    ends with the mesh offsets (uint64 for v4, uint32 for v3) and a uint32 mesh count."""
    import struct
    import zlib

    data = open(path, "rb").read()
    count = struct.unpack("<I", data[-4:])[0]
    if not 0 <= shape_index < count:
        raise ValueError(f"{path} holds {count} meshes, shape_index {shape_index} asked")
    version = struct.unpack("<HH", data[:4])[1]
    if version == 4:
        offsets = struct.unpack(f"<{count}Q", data[-4 - 8 * count:-4])
    elif version == 3:
        offsets = struct.unpack(f"<{count}I", data[-4 - 4 * count:-4])
    else:
        raise ValueError(f"unsupported .serialized version {version}")
    start = offsets[shape_index]
    magic, ver = struct.unpack("<HH", data[start:start + 4])
    if magic != 0x041C:
        raise ValueError("not a Mitsuba .serialized mesh")
    raw = zlib.decompressobj().decompress(data[start + 4:])
    pos = 0
    flags = struct.unpack_from("<I", raw, pos)[0]; pos += 4
    if ver == 4:
        end = raw.index(b"\0", pos)
        pos = end + 1
    nv, nt = struct.unpack_from("<QQ", raw, pos); pos += 16
    dt = np.dtype("<f8") if flags & 0x2000 else np.dtype("<f4")

    def take(n, dtype):
        nonlocal pos
        a = np.frombuffer(raw, dtype=dtype, count=n, offset=pos)
        pos += n * dtype.itemsize
        return a

    verts = take(3 * nv, dt).reshape(nv, 3).astype(np.float64)
    normals = take(3 * nv, dt).reshape(nv, 3).astype(np.float64) if flags & 0x0001 else None
    if flags & 0x0002:
        take(2 * nv, dt)
    if flags & 0x0008:
        take(3 * nv, dt)
    faces = take(3 * nt, np.dtype("<u4")).reshape(nt, 3).astype(np.int64)
    if flags & 0x0010:  # face normals requested by the file: ignore the stored vertex normals
        normals = None
    return verts, faces, normals


def triangles(vertices: np.ndarray, faces: np.ndarray, to_world: np.ndarray, material_index: int,
              vertex_normals: np.ndarray = None, uvs: np.ndarray = None, flip_tex_coords: bool = True):
    """(F,16) triangle records in world space; degenerate triangles are dropped.
    vertex_normals: (V,3) per vertex or (F,3,3) per face corner; uvs: (V,2) or (F,3,2) likewise.
    Without either it returns the records; with normals only (records, (F,9) world-space unit
    normals); with texture coordinates (records, normals or None, (F,6) uv0 uv1 uv2).
    flip_tex_coords: v -> 1 - v, the default of Mitsuba's `obj` plugin (image row 0 is then v = 0)."""
    m = np.asarray(to_world, np.float64)
    faces = np.asarray(faces, np.int64)
    v = np.asarray(vertices, np.float64) @ m[:3, :3].T + m[:3, 3]
    v = v.astype(np.float32)  # the kernels see fp32 vertices: derive everything from those
    a, b, c = v[faces[:, 0]], v[faces[:, 1]], v[faces[:, 2]]
    e1, e2 = (b - a).astype(np.float32), (c - a).astype(np.float32)
    n = np.cross(e1.astype(np.float64), e2.astype(np.float64))
    ln = np.linalg.norm(n, axis=1)
    keep = ln > 0
    out = np.zeros((int(keep.sum()), TRI_STRIDE), np.float32)
    out[:, 0:3], out[:, 3:6], out[:, 6:9] = a[keep], e1[keep], e2[keep]
    out[:, 9:12] = (n[keep] / ln[keep, None]).astype(np.float32)
    out[:, 12] = np.float32(material_index)
    if vertex_normals is None and uvs is None:
        return out
    fn = None
    if vertex_normals is not None:
        vn = np.asarray(vertex_normals, np.float64)
        nw = vn.reshape(-1, 3) @ np.linalg.inv(m[:3, :3])  # normals transform by the inverse transpose
        nl = np.linalg.norm(nw, axis=1)
        nw = np.where(nl[:, None] > 0, nw / np.maximum(nl[:, None], 1e-300), 0.0)
        corner = nw.reshape(-1, 3, 3) if vn.ndim == 3 else nw[faces]     # (F,3,3)
        fn = corner.reshape(-1, 9)[keep].astype(np.float32)
    if uvs is None:
        return out, fn
    uv = np.asarray(uvs, np.float64)
    corner = (uv if uv.ndim == 3 else uv[faces]).astype(np.float32)      # (F,3,2)
    if flip_tex_coords:
        corner = corner.copy()
        corner[:, :, 1] = np.float32(1.0) - corner[:, :, 1]
    return out, fn, corner.reshape(-1, 6)[keep].astype(np.float32)


def build_bvh(tris: np.ndarray, per_triangle: np.ndarray = None):
    """Four-wide BVH over triangle records (a binary tree built with the surface-area heuristic,
    <= MAX_LEAF triangles per leaf, then collapsed); returns (nodes (M,32) uint32 bit patterns,
    triangles in leaf order); with `per_triangle` (T, k) data also that array in the same order."""
    tris = np.ascontiguousarray(tris, np.float32).reshape(-1, TRI_STRIDE)
    n = tris.shape[0]
    if n == 0:
        empty = np.zeros((0, BVH_STRIDE), np.uint32)
        return (empty, tris) if per_triangle is None else (empty, tris, per_triangle)
    v0, v1, v2 = tris[:, 0:3], tris[:, 0:3] + tris[:, 3:6], tris[:, 0:3] + tris[:, 6:9]
    lo = np.minimum(np.minimum(v0, v1), v2)
    hi = np.maximum(np.maximum(v0, v1), v2)
    lo64, hi64 = lo.astype(np.float64), hi.astype(np.float64)
    cen = 0.5 * (lo64 + hi64)
    order = np.arange(n)
    nodes: List[List[int]] = [[0, 0]]
    bounds: List[Tuple[np.ndarray, np.ndarray]] = [(None, None)]

    def half_area(a, b):
        e = b - a
        return e[..., 0] * e[..., 1] + e[..., 1] * e[..., 2] + e[..., 2] * e[..., 0]

    def sah_split(idx):
        """Full-sweep surface-area heuristic over the three axes: (cost, axis, sorted indices, position)."""
        best = None
        cnt = idx.shape[0]
        k = np.arange(1, cnt, dtype=np.float64)
        for axis in range(3):
            srt = idx[np.argsort(cen[idx, axis], kind="stable")]
            l, h = lo64[srt], hi64[srt]
            left = half_area(np.minimum.accumulate(l, axis=0), np.maximum.accumulate(h, axis=0))[:-1]
            right = half_area(np.minimum.accumulate(l[::-1], axis=0)[::-1], np.maximum.accumulate(h[::-1], axis=0)[::-1])[1:]
            cost = left * k + right * (cnt - k)
            pos = int(np.argmin(cost))
            if best is None or cost[pos] < best[0]:
                best = (float(cost[pos]), axis, srt, pos + 1)
        return best

    # The binary tree first.  Big nodes split by the surface-area heuristic, small ones (and very
    # deep ones, to bound the depth) at the median of the widest axis.
    todo = [(0, 0, n, 0)]
    while todo:
        me, first, count, depth = todo.pop()
        idx = order[first:first + count]
        bounds[me] = (lo[idx].min(axis=0), hi[idx].max(axis=0))
        if count <= MAX_LEAF:
            nodes[me] = [first, LEAF_FLAG | count]
            continue
        if count > 2 * MAX_LEAF and depth < 40:
            _, axis, srt, mid = sah_split(idx)
            order[first:first + count] = srt
        else:
            c = cen[idx]
            ext = c.max(axis=0) - c.min(axis=0)
            axis = int(np.argmax(ext))
            if ext[axis] != 0.0:  # (all centroids coincide: split by position in the list)
                order[first:first + count] = idx[np.argsort(c[:, axis], kind="stable")]
            mid = count // 2
        left = len(nodes)
        nodes += [[0, 0], [0, 0]]
        bounds += [(None, None), (None, None)]
        nodes[me] = [left, (left + 1) | (axis << 29)]
        todo.append((left + 1, first + mid, count - mid, depth + 1))
        todo.append((left, first, mid, depth + 1))
    # Collapse to four children per node: a node adopts the children of its (by surface area) biggest
    # inner child until it has four -- half the dependent steps of a walk.  Leaves are referenced
    # straight from their parent.  Numbering: the first BVH_HOT_NODES numbers go to the nodes with the
    # biggest boxes -- the ones most rays open; the ray-casting kernels keep the first few dozen nodes
    # of the table in LDS -- and the rest are numbered depth first, a subtree's nodes together.  Either
    # way a node is numbered after its parent.
    def is_leaf(b):
        return bool(nodes[b][1] & LEAF_FLAG)

    def area(b):
        return float(half_area(bounds[b][0].astype(np.float64), bounds[b][1].astype(np.float64)))

    sequence: List[int] = []                  # binary nodes that become wide nodes, in numbering order
    kids_of = {}
    frontier = [] if is_leaf(0) else [0]
    if is_leaf(0):
        sequence, kids_of = [0], {0: [0]}     # a mesh of <= MAX_LEAF triangles: one node with one leaf child
    while frontier:
        if len(sequence) < BVH_HOT_NODES:
            b = frontier.pop(max(range(len(frontier)), key=lambda q: (area(frontier[q]), q)))
        else:
            b = frontier.pop()
        kids = [nodes[b][0], nodes[b][1] & 0x1FFFFFFF]
        while len(kids) < BVH_WIDTH:
            inner = [k for k in kids if not is_leaf(k)]
            if not inner:
                break
            big = max(inner, key=area)
            at = kids.index(big)
            kids[at:at + 1] = [nodes[big][0], nodes[big][1] & 0x1FFFFFFF]
        sequence.append(b)
        kids_of[b] = kids
        for k in reversed(kids):  # (a stack: the first child's subtree is numbered first)
            if not is_leaf(k):
                frontier.append(k)
    ids = {b: i for i, b in enumerate(sequence)}
    wide = [kids_of[b] for b in sequence]
    out = np.zeros((len(wide), BVH_STRIDE), np.uint32)
    inf = np.float32(np.inf)
    for i, kids in enumerate(wide):
        box = np.empty((6, BVH_WIDTH), np.float32)
        box[0:3], box[3:6] = inf, -inf
        out[i, 24:28] = EMPTY_CHILD
        for k, b in enumerate(kids):
            box[0:3, k], box[3:6, k] = bounds[b][0], bounds[b][1]
            if is_leaf(b):
                first, count = nodes[b][0], nodes[b][1] & 0x7FFFFFFF
                out[i, 24 + k] = LEAF_FLAG | ((count - 1) << 28) | first
            else:
                out[i, 24 + k] = ids[b]
        out[i, 0:24] = box.reshape(-1).view(np.uint32)
        out[i, 28] = len(kids)
    if per_triangle is None:
        return out, np.ascontiguousarray(tris[order])
    return out, np.ascontiguousarray(tris[order]), np.ascontiguousarray(np.asarray(per_triangle)[order])


def icosphere(subdivisions: int = 2) -> Tuple[np.ndarray, np.ndarray]:
    """Unit icosphere (vertices, faces): a closed test mesh."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
         (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6),
         (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7),
         (9, 8, 1)]
    verts = [np.asarray(p, np.float64) / np.linalg.norm(p) for p in v]
    faces = [tuple(x) for x in f]
    for _ in range(subdivisions):
        cache = {}

        def mid(i, j):
            key = (min(i, j), max(i, j))
            if key not in cache:
                m = verts[i] + verts[j]
                verts.append(m / np.linalg.norm(m))
                cache[key] = len(verts) - 1
            return cache[key]

        nf = []
        for a, b, c in faces:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        faces = nf
    return np.asarray(verts), np.asarray(faces, np.int64)
