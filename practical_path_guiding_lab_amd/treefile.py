"""Host-side readers of a saved SD-tree: the part of KDTreeNode / QuadTreeNode that the reference's
offline tools use on a file written by saveSDTreeToFile (tree_plotter.py:25-30, 159-163: `loadFromFile`;
:38 `getAllLeafNodeIndex(rootIndex)`; :56 `getBBox`; src/kdtree.py:53-76, src/quadtree.py:58-85, 288-345).

Plain numpy over the 23-key npz: files written by this library (SDTree.saveToFile)
and files written by the reference load alike.  Not on the hot path; the trees a render uses live in
device memory behind the C ABI (sdtree.SDTree).
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np


def _boxes(bmin, bmax, n: int, k: int) -> Tuple[np.ndarray, np.ndarray]:
    """The (n, k) bbox_min / bbox_max columns of a saved tree.  The on-disk orientation is the reference's:
    (n, k), one row per node, as Dr.Jit's `.numpy()` gives a Vector column (kdtree.py:575-602) and as
    SDTree.saveToFile writes it.  A (k, n) file (a column-major writer) is accepted when its shape says so
    unambiguously.  With n == k -- a 3-node KD tree, a 2-node quadtree forest -- the shape says nothing: the
    reading is chosen by what every saved tree satisfies -- no box is inverted and every node's box lies
    inside node 0's (the KD root holds every node; every quadtree node lies in the unit square that root 0
    is).  The reference's orientation is taken when it passes (it is the only writer known), the transposed
    one when only that passes, and a file that passes neither way is refused rather than read wrongly."""
    bmin, bmax = np.asarray(bmin, np.float32), np.asarray(bmax, np.float32)
    if bmin.shape != bmax.shape:
        raise ValueError(f"bbox_min {bmin.shape} and bbox_max {bmax.shape} differ in shape")
    if n != k:
        if bmin.shape == (n, k):
            return bmin, bmax
        if bmin.shape == (k, n):
            return bmin.T, bmax.T
        raise ValueError(f"expected a column of {n} vectors of {k}, got {bmin.shape}")
    if bmin.shape != (n, k):
        raise ValueError(f"expected a column of {n} vectors of {k}, got {bmin.shape}")

    def plausible(lo, hi):
        return bool(np.all(lo <= hi) and np.all(lo >= lo[0]) and np.all(hi <= hi[0]))

    if plausible(bmin, bmax):
        return bmin, bmax
    if plausible(bmin.T, bmax.T):
        return bmin.T, bmax.T
    raise ValueError(f"{n} nodes of {k} coordinates: neither reading of the square bbox columns is a tree (boxes inside node 0's)")


class KDTreeNode:
    """src/kdtree.py:16-76."""

    def loadFromFile(self, dataNumpy) -> None:  # src/kdtree.py:53-63
        self.depth = np.asarray(dataNumpy["kdtree_depth"], np.uint32)
        n = self.depth.shape[0]
        self.bbox_min, self.bbox_max = _boxes(dataNumpy["kdtree_bbox_min"], dataNumpy["kdtree_bbox_max"], n, 3)
        self.vertCount = np.asarray(dataNumpy["kdtree_vertCount"], np.float32)
        self.isLeaf = np.asarray(dataNumpy["kdtree_isLeaf"], bool)
        self.quadTreeRootIndex = np.asarray(dataNumpy["kdtree_quadTreeRootIndex"], np.uint32)
        self.child_left_index = np.asarray(dataNumpy["kdtree_child_left_index"], np.uint32)
        self.child_right_index = np.asarray(dataNumpy["kdtree_child_right_index"], np.uint32)

    def getWidth(self) -> int:  # :66
        return int(self.depth.shape[0])

    def getBBox(self, idx) -> Tuple[np.ndarray, np.ndarray]:  # :70-76 (min, max) of the nodes `idx`
        idx = np.asarray(idx, np.int64)
        return self.bbox_min[idx], self.bbox_max[idx]

    def getAllLeafNodeIndex(self) -> np.ndarray:  # KDTree.getAllLeafNodeIndex, :173-177
        return np.nonzero(self.isLeaf)[0].astype(np.uint32)

    def getLeafNodeIndex(self, position) -> np.ndarray:
        """KDTree.getLeafNodeIndex (:435-470) on the host, for the plotter's findLeafNode
        (tree_plotter.py:170-): positions (n, 3); a point outside the root box stays at node 0."""
        p = np.asarray(position, np.float32).reshape(-1, 3)
        node = np.zeros(p.shape[0], np.int64)
        active = np.all((p >= self.bbox_min[0]) & (p <= self.bbox_max[0]), axis=1)
        for _ in range(64):
            active &= ~self.isLeaf[node]
            if not active.any():
                break
            left, right = self.child_left_index[node].astype(np.int64), self.child_right_index[node].astype(np.int64)
            in_left = np.all((p >= self.bbox_min[left]) & (p <= self.bbox_max[left]), axis=1)
            in_right = np.all((p >= self.bbox_min[right]) & (p <= self.bbox_max[right]), axis=1)
            nxt = np.where(in_left, left, node)
            nxt = np.where(in_right, right, nxt)  # (the right child wins a tie on the split plane, :462-468)
            node = np.where(active, nxt, node)
        return node.astype(np.uint32)


class QuadTreeNode:
    """src/quadtree.py:12-85, 288-345."""

    def loadFromFile(self, dataNumpy) -> None:  # src/quadtree.py:58-71
        self.rootNodeIndex = np.asarray(dataNumpy["quadtree_rootNodeIndex"], np.uint32)
        self.depth = np.asarray(dataNumpy["quadtree_depth"], np.uint32)
        n = self.depth.shape[0]
        self.bbox_min, self.bbox_max = _boxes(dataNumpy["quadtree_bbox_min"], dataNumpy["quadtree_bbox_max"], n, 2)
        self.irradiance = np.asarray(dataNumpy["quadtree_irradiance"], np.float32)
        self.isLeaf = np.asarray(dataNumpy["quadtree_isLeaf"], bool)
        self.refinementThreshold = np.asarray(dataNumpy["quadtree_refinementThreshold"], np.float32)
        self.child_1_index = np.asarray(dataNumpy["quadtree_child_1_index"], np.uint32)
        self.child_2_index = np.asarray(dataNumpy["quadtree_child_2_index"], np.uint32)
        self.child_3_index = np.asarray(dataNumpy["quadtree_child_3_index"], np.uint32)
        self.child_4_index = np.asarray(dataNumpy["quadtree_child_4_index"], np.uint32)

    def getWidth(self) -> int:  # :74
        return int(self.depth.shape[0])

    def getBBox(self, idx) -> Tuple[np.ndarray, np.ndarray]:  # :78-85
        idx = np.asarray(idx, np.int64)
        return self.bbox_min[idx], self.bbox_max[idx]

    def getAllLeafNodeIndex(self, rootIndex: Optional[np.ndarray] = None) -> np.ndarray:
        """:288-345: every leaf of every tree, or the leaves below the roots `rootIndex` (tree numbers) in
        the reference's order -- depth by depth, a depth's frontier being all first children of the previous
        depth's inner nodes, then all second, third and fourth children."""
        if rootIndex is None or np.size(rootIndex) == 0:
            return np.nonzero(self.isLeaf)[0].astype(np.uint32)
        node = self.rootNodeIndex[np.asarray(rootIndex, np.int64).reshape(-1)].astype(np.int64)
        out = []
        while node.size:
            leaf = self.isLeaf[node]
            out.append(node[leaf])
            inner = node[~leaf]
            node = np.concatenate([self.child_1_index[inner], self.child_2_index[inner], self.child_3_index[inner],
                                   self.child_4_index[inner]]).astype(np.int64)
        return np.concatenate(out).astype(np.uint32) if out else np.zeros(0, np.uint32)

    def sampleIrradiance(self, rootIndex, position) -> np.ndarray:
        """QuadTreePlotter.sampleIrradiance (tree_plotter.py:45-100): the irradiance of the leaf of tree
        `rootIndex[i]` that holds canonical position[i] (later children win ties, as the splat and the
        reference's sequential selects, quadtree.py:424-438); 0 outside the root cell."""
        pos = np.asarray(position, np.float32).reshape(-1, 2)
        node = self.rootNodeIndex[np.asarray(rootIndex, np.int64).reshape(-1)].astype(np.int64)
        active = np.all((pos >= self.bbox_min[node]) & (pos <= self.bbox_max[node]), axis=1)
        inside = active.copy()
        for _ in range(64):
            active &= ~self.isLeaf[node]
            if not active.any():
                break
            nxt = node.copy()
            for ch in (self.child_1_index, self.child_2_index, self.child_3_index, self.child_4_index):
                c = ch[node].astype(np.int64)
                hit = np.all((pos >= self.bbox_min[c]) & (pos <= self.bbox_max[c]), axis=1)
                nxt = np.where(hit, c, nxt)
            node = np.where(active, nxt, node)
        return np.where(inside, self.irradiance[node], np.float32(0.0)).astype(np.float32)

    def getMaxDepth(self, rootIndex: int) -> int:
        """QuadTreePlotter.getMaxDepth (tree_plotter.py:33-42)."""
        return int(self.depth[self.getAllLeafNodeIndex(np.array([rootIndex]))].max())


def load(fileName: str) -> Tuple[KDTreeNode, QuadTreeNode]:
    """KDTreePlotter.__init__ (tree_plotter.py:153-166): both node tables of one saved SD-tree."""
    data = np.load(fileName)
    kd, qt = KDTreeNode(), QuadTreeNode()
    kd.loadFromFile(data)
    qt.loadFromFile(data)
    return kd, qt
