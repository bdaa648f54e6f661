"""A SECOND restatement of the reference's SD-tree queries, for pinning the C oracle against something that is not itself.

oracle/pg_oracle.c walks one lane at a time (a scalar `while`).  The reference is a WAVEFRONT program: all lanes advance
together under masks, a masked gather yields 0 for an inactive lane, a masked scatter `x[mask] = v` overwrites in program order.
This module restates the same functions in that form -- numpy arrays, explicit masks, the reference's statement order -- on the
reference's own column schema (the 23 npz keys), so that a slip in the scalar restatement (a tie rule, an early exit, a draw too
many) shows up as a difference between two independently structured programs.  Test infrastructure only; float32 throughout,
every operation rounded by itself (numpy does not contract); the transcendental maps and the sampler are the contract's
(oracle.pg_oracle: dir_to_canonical, canonical_to_dir, rng_next_f32), because those are defined by DESIGN.md 4, not by the tree.

  kd_get_leaf_node_index   src/kdtree.py:435-470
  quad_pdf                 src/quadtree.py:1001-1101
  quad_sample              src/quadtree.py:931-998
  quad_add_propagate       src/quadtree.py:398-441 (one addIrradiancePropagate: which nodes a record adds to, in order)
"""
import numpy as np

from oracle import pg_oracle as po

F = np.float32
INV_FOUR_PI = F(0.07957747154594767)


def _gather(col, idx, mask):
    """dr.gather(col, idx, mask): the value where the mask is set, 0 elsewhere."""
    out = np.zeros(idx.shape + col.shape[1:], col.dtype)
    out[mask] = col[idx[mask]]
    return out


def _contains(bmin, bmax, idx, p):
    """BoundingBox.contains of box `idx` for points p (n, d): inclusive on every face; NaN fails."""
    lo, hi = bmin[idx], bmax[idx]
    return np.all((p >= lo) & (p <= hi), axis=1)


def kd_get_leaf_node_index(t, p, active=None):
    n = p.shape[0]
    node = np.zeros(n, np.uint32)
    search = _contains(t["kdtree_bbox_min"], t["kdtree_bbox_max"], np.zeros(n, np.int64), p)
    if active is not None:
        search &= active.astype(bool)
    for _ in range(64):
        if not search.any():
            break
        is_leaf = _gather(t["kdtree_isLeaf"], node, search)
        search = search & ~is_leaf
        left = _gather(t["kdtree_child_left_index"], node, search)
        right = _gather(t["kdtree_child_right_index"], node, search)
        in_left = _contains(t["kdtree_bbox_min"], t["kdtree_bbox_max"], left, p)
        node[in_left & search] = left[in_left & search]
        in_right = _contains(t["kdtree_bbox_min"], t["kdtree_bbox_max"], right, p)
        node[in_right & search] = right[in_right & search]      # (the later statement wins a tie)
    return node


def _children(t, node, mask):
    return [_gather(t["quadtree_child_%d_index" % k], node, mask) for k in (1, 2, 3, 4)]


def quad_pdf(t, root_index, direction, active=None):
    n = root_index.shape[0]
    node = t["quadtree_rootNodeIndex"][root_index].astype(np.uint32)
    pdf = np.ones(n, F)
    act = np.ones(n, bool) if active is None else active.astype(bool).copy()
    pos = po.dir_to_canonical(np.ascontiguousarray(direction.T)).T.astype(F)     # (n, 2)
    irr, bmin, bmax = t["quadtree_irradiance"], t["quadtree_bbox_min"], t["quadtree_bbox_max"]
    for _ in range(64):
        if not act.any():
            break
        is_leaf = _gather(t["quadtree_isLeaf"], node, act)
        m = act & is_leaf
        pdf[m] = pdf[m] * INV_FOUR_PI
        act = act & ~is_leaf
        ch = _children(t, node, act)
        inside = [_contains(bmin, bmax, c, pos) for c in ch]
        node_irr = _gather(irr, node, act)
        c_irr = [_gather(irr, c, act & ins) for c, ins in zip(ch, inside)]
        child_irr = np.where(inside[0], c_irr[0], np.where(inside[1], c_irr[1], np.where(inside[2], c_irr[2],
                             np.where(inside[3], c_irr[3], F(0)))))           # (the FIRST containing child's energy)
        with np.errstate(all="ignore"):
            step = (F(4) * child_irr) / node_irr
            pdf[act] = (pdf * step)[act]
        nan = np.isnan(pdf) & act
        pdf[nan] = F(0)
        act = act & ~nan
        for c, ins in zip(ch, inside):                                          # (the LAST containing child is walked into)
            node[act & ins] = c[act & ins]
    return pdf


def quad_sample(t, root_index, state, inc, active=None):
    """Returns (directions (n, 3), the sampled canonical positions (n, 2)); `state` is advanced in place: next_2d then next_1d
    for every lane that entered the iteration (Dr.Jit masks a loop body's side effects with the loop's entry mask)."""
    n = root_index.shape[0]
    node = t["quadtree_rootNodeIndex"][root_index].astype(np.uint32)
    pos = np.zeros((n, 2), F)
    act = np.ones(n, bool) if active is None else active.astype(bool).copy()
    irr, bmin, bmax = t["quadtree_irradiance"], t["quadtree_bbox_min"], t["quadtree_bbox_max"]

    def draw(mask):
        out = np.zeros(n, F)
        if mask.any():
            st = np.ascontiguousarray(state[mask])
            out[mask] = po.rng_next_f32(st, np.ascontiguousarray(inc[mask]))
            state[mask] = st
        return out

    for _ in range(64):
        if not act.any():
            break
        entered = act.copy()
        is_leaf = _gather(t["quadtree_isLeaf"], node, act)
        lo, hi = _gather(bmin, node, act), _gather(bmax, node, act)
        u = np.stack([draw(entered), draw(entered)], axis=1)                   # next_2d
        m = act & is_leaf
        pos[m] = (lo + u * (hi - lo))[m]
        act = act & ~is_leaf
        ch = _children(t, node, act)
        e = [_gather(irr, c, act) for c in ch]
        c1 = e[0]
        c2 = e[1] + c1
        c3 = e[2] + c2
        c4 = e[3] + c3
        s = draw(entered) * c4                                                  # next_1d
        pick = [s < c1, (c1 <= s) & (s < c2), (c2 <= s) & (s < c3), c3 <= s]
        for c, pk in zip(ch, pick):
            node[act & pk] = c[act & pk]
        # (a lane none of whose bins holds s -- NaN energies -- would spin for ever in the reference; the oracle stops it)
        act = act & (pick[0] | pick[1] | pick[2] | pick[3])
    return po.canonical_to_dir(np.ascontiguousarray(pos.T)).T, pos


def quad_add_propagate(t, root_index, position):
    """The nodes one addIrradiancePropagate call adds to, lane by lane: a list of (lane indices, node indices) per loop
    iteration, in order."""
    n = root_index.shape[0]
    node = t["quadtree_rootNodeIndex"][root_index].astype(np.uint32)
    bmin, bmax = t["quadtree_bbox_min"], t["quadtree_bbox_max"]
    act = _contains(bmin, bmax, node, position)
    out = []
    for _ in range(64):
        if not act.any():
            break
        out.append((np.nonzero(act)[0], node[act].copy()))
        is_leaf = _gather(t["quadtree_isLeaf"], node, act)
        act = act & ~is_leaf
        for c in _children(t, node, act):
            ins = _contains(bmin, bmax, c, position)
            node[ins & act] = c[ins & act]
    return out
