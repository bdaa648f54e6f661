"""The ordering step of a sorted bounce (csrc/pg_sort.hip, behind pg_sort_places): two unstable counting passes written for
the renderer's 16-bit key.  What the renderer relies on, and nothing more: the first `live` entries of the result are a
PERMUTATION of the live places (every path is shaded once, no thread gets a place without a path), in the order of the keys'
high byte, and inside a high byte in the order of the low byte up to one tile of the second pass.  The case that broke a
work-in-progress build (a torus pass: rays that left the scene, key 0xfffe, behind them the empty tail of the list, key
0xffff) is here by name."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tree():
    from practical_path_guiding_lab_amd.sdtree import SDTree
    return SDTree(0)


def _check(tree, keys_np, live):
    import torch
    n = keys_np.shape[0]
    keys = torch.from_numpy(keys_np.astype(np.uint16).view(np.int16)).cuda()
    out = tree.sortPlaces(keys, live).cpu().numpy().view(np.uint32)
    L = n if live is None else min(n, live)
    head = out[:L].astype(np.int64)
    assert head.min(initial=0) >= 0 and head.max(initial=0) < max(L, 1)
    assert np.array_equal(np.sort(head), np.arange(L)), "not a permutation of the live places"
    if live is not None and L < n:
        assert (out[L:] == 0xFFFFFFFF).all()       # behind the live places nothing is written
    k = keys_np[head].astype(np.int64)
    assert (np.diff(k >> 8) >= 0).all(), "high bytes out of order"
    return k


@pytest.mark.parametrize("n", [1, 2, 255, 4096, 4097, 70001, 1_000_003, 8_619_131])
def test_places_are_a_permutation_in_high_byte_order(tree, n):
    rng = np.random.default_rng(n)
    keys = rng.integers(0, 0xFFFE, n, dtype=np.int64)
    k = _check(tree, keys, None)
    # what include/pgsd.h promises about the LOW byte depends on n: a tile of the second pass (4096 consecutive pairs of a sequence
    # sorted by the low byte) spans about 2^20 / n of its 256 values -- inside one high byte the low byte never steps DOWN by more
    same = np.diff(k >> 8) == 0
    if n >= (1 << 21):   # long lists: at most two neighbouring low bytes per tile
        assert (np.diff(k & 255)[same] >= -1).all()
    # (shorter lists: coarser, and below 2^20 places no promise about the low byte at all -- high-byte order only, asserted by _check)


def test_rays_that_left_the_scene_and_the_empty_tail(tree):
    """Live places with key 0xfffe (the ray left the scene) and, behind the live count, places without a path (0xffff): no
    empty place may come to stand among the first `live` entries whatever the passes do inside a tile."""
    rng = np.random.default_rng(5)
    n, live = 3_000_001, 2_345_678
    keys = rng.integers(0, 0xFFFE, n, dtype=np.int64)
    keys[rng.random(n) < 0.3] = 0xFFFE
    keys[rng.random(n) < 0.2] = 0xFFFD
    keys[live:] = 0xFFFF
    _check(tree, keys, live)
    _check(tree, keys, n + 5)          # a live count beyond the list: all n places
    keys[:] = 0x1234                    # one cell for everybody
    _check(tree, keys, 1_000_000)
    _check(tree, np.arange(n, dtype=np.int64) * 65536 // n, None)   # already in order


def test_sort_places_refuses_bad_arguments(tree):
    import ctypes as C
    import torch
    from practical_path_guiding_lab_amd import _native as N
    L = N.lib()
    k = torch.zeros(16, dtype=torch.int16, device="cuda")
    o = torch.zeros(16, dtype=torch.int32, device="cuda")
    assert L.pg_sort_places(tree._h, 16, None, None, o.data_ptr(), None) < 0
    assert L.pg_sort_places(tree._h, 16, k.data_ptr(), None, None, None) < 0
    assert L.pg_sort_places(tree._h, 1 << 28, k.data_ptr(), None, o.data_ptr(), None) < 0
    assert L.pg_sort_places(tree._h, 0, None, None, None, None) == 0
