"""Pins the oracle's scalar arithmetic (oracle/pgo_math.h) with hand-derivable and
exactly computable cases.  CPU only."""
from fractions import Fraction

import numpy as np


def test_dir_canonical_known_answer(oracle):
    # reference self-test src/common.py:274-279: dir (0,1,0) -> (phi=0.25, 0.5) -> (0,1,0)
    c = oracle.dir_to_canonical(np.array([[0.0], [1.0], [0.0]], np.float32))
    assert c[0, 0] == np.float32(0.25) and c[1, 0] == np.float32(0.5)
    d = oracle.canonical_to_dir(c)
    assert abs(d[0, 0]) < 1e-7 and d[1, 0] == np.float32(1.0) and d[2, 0] == 0.0
    # axis cases derivable by hand: +x -> phi 0; -x -> 0.5; -y -> 0.75; +z -> cos 1; -z -> 0
    dirs = np.array([[1, 0, 0], [-1, 0, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], np.float32).T
    c = oracle.dir_to_canonical(dirs)
    np.testing.assert_array_equal(c[0], np.array([0, 0.5, 0.75, 0, 0], np.float32))
    np.testing.assert_array_equal(c[1], np.array([0.5, 0.5, 0.5, 1.0, 0.0], np.float32))


def test_dir_to_canonical_nonfinite_is_origin(oracle):
    # src/common.py:156-158
    dirs = np.array([[np.nan, 0, 0], [0, np.inf, 0], [0, 0, -np.inf]], np.float32).T
    np.testing.assert_array_equal(oracle.dir_to_canonical(dirs), np.zeros((2, 3), np.float32))


def test_dir_to_canonical_range_and_wrap(oracle):
    rng = np.random.default_rng(1)
    d = rng.normal(size=(3, 20000)).astype(np.float32)
    d /= np.linalg.norm(d, axis=0, keepdims=True)
    c = oracle.dir_to_canonical(d)
    assert (c >= 0).all() and (c <= 1).all()
    back = oracle.canonical_to_dir(c)
    # sinTheta = sqrt(1 - cos^2) cancels near the poles (the reference's own formula): ~1e-5 there
    assert np.abs(back - d).max() < 5e-5
    # tiny negative phi wraps to exactly two_pi -> x == 1.0 (still inside the root square)
    c = oracle.dir_to_canonical(np.array([[1.0], [-1e-12], [0.0]], np.float32))
    assert c[0, 0] == np.float32(1.0)


def test_sincos_atan2_correctly_rounded(oracle):
    rng = np.random.default_rng(2)
    phi = np.concatenate([rng.uniform(0, 2 * np.pi, 50000), [0.0, np.pi / 2, np.pi, 2 * np.pi]]).astype(np.float32)
    s, c = oracle.sincos(phi)
    np.testing.assert_array_equal(s, np.sin(phi.astype(np.float64)).astype(np.float32))
    np.testing.assert_array_equal(c, np.cos(phi.astype(np.float64)).astype(np.float32))
    y = rng.normal(size=50000).astype(np.float32)
    x = rng.normal(size=50000).astype(np.float32)
    a = oracle.atan2(y, x)
    np.testing.assert_array_equal(a, np.arctan2(y.astype(np.float64), x.astype(np.float64)).astype(np.float32))
    # IEEE zero/sign cases
    yy = np.array([0.0, -0.0, 0.0, -0.0, 1.0, -1.0, 0.0], np.float32)
    xx = np.array([-0.0, -0.0, 0.0, 0.0, 0.0, 0.0, -2.0], np.float32)
    np.testing.assert_array_equal(oracle.atan2(yy, xx), np.arctan2(yy, xx).astype(np.float32))


def test_exp_log_erf_erfinv_contract(oracle):
    """pgo_math.h's exp/log (double series, rounded once: within half an ulp of the true value on
    these samples) and erf/erfinv (A&S 7.1.26, Giles: ~1.5e-7) -- the functions the rough-conductor
    BSDF of the substrate is built from."""
    from scipy import special as sp

    rng = np.random.default_rng(12)
    x = rng.uniform(-87, 88, 200000).astype(np.float32)
    np.testing.assert_array_equal(oracle.math1("exp", x), np.exp(x.astype(np.float64)).astype(np.float32))
    x = np.exp(rng.uniform(-87, 88, 200000)).astype(np.float32)
    np.testing.assert_array_equal(oracle.math1("log", x), np.log(x.astype(np.float64)).astype(np.float32))
    x = rng.uniform(-5, 5, 200000).astype(np.float32)
    assert np.abs(oracle.math1("erf", x) - sp.erf(x.astype(np.float64))).max() < 2.5e-7
    x = rng.uniform(-1, 1, 200000).astype(np.float32)
    r = sp.erfinv(x.astype(np.float64))
    assert (np.abs(oracle.math1("erfinv", x) - r) / np.maximum(np.abs(r), 1e-20)).max() < 3e-7
    # special values
    with np.errstate(all="ignore"):
        e = oracle.math1("exp", [np.inf, -np.inf, 0.0, -100.0, 89.0, -87.5])
        assert e[0] == np.inf and e[1] == 0 and e[2] == 1 and e[3] == 0 and e[4] == np.inf and e[5] == 0
        l = oracle.math1("log", [0.0, -1.0, np.inf, 1.0, 1e-45])
        assert l[0] == -np.inf and np.isnan(l[1]) and l[2] == np.inf and l[3] == 0
        assert l[4] == np.float32(np.log(float(np.float32(1e-45))))
        assert list(oracle.math1("erf", [np.inf, -np.inf, 10.0])) == [1.0, -1.0, 1.0]
        v = oracle.math1("erfinv", [1.0, -1.0, 0.0, 2.0])
        assert v[0] == np.inf and v[1] == -np.inf and v[2] == 0 and np.isnan(v[3])


def _py_quantize(w: np.float32) -> int:
    if np.isnan(w):
        return 0
    lim = Fraction(2) ** 48
    if np.isinf(w):
        f = lim if w > 0 else -lim
    else:
        f = Fraction(float(w))
        if abs(f) >= lim:
            f = lim if f > 0 else -lim
    q = f * (1 << 40)
    n = abs(q.numerator) // q.denominator  # truncation toward zero
    return -n if q < 0 else n


def _py_to_f32(v: int) -> np.float32:
    # exact integer -> nearest-even fp32, then exact 2^-40 scaling
    if v == 0:
        return np.float32(0)
    neg, mag = v < 0, abs(v)
    msb = mag.bit_length() - 1
    if msb <= 23:
        mant, sh = mag, 0
    else:
        sh = msb - 23
        mant, rem = mag >> sh, mag & ((1 << sh) - 1)
        half = 1 << (sh - 1)
        if rem > half or (rem == half and (mant & 1)):
            mant += 1
    f = np.float32(np.ldexp(np.float64(mant), sh - 40))
    return -f if neg else f


def test_quantize_and_accumulator_to_float_exact(oracle):
    rng = np.random.default_rng(3)
    w = np.concatenate([
        np.ldexp(rng.uniform(0.5, 1, 4000), rng.integers(-60, 60, 4000)) * rng.choice([-1, 1], 4000),
        [0.0, -0.0, np.inf, -np.inf, np.nan, 2.0 ** 48, 2.0 ** 47, 2.0 ** -40, 2.0 ** -41, 1e-45, 3.0e38],
    ]).astype(np.float32)
    lo, hi = oracle.quantize(w)
    got = [(int(h) << 64) + int(l) for l, h in zip(lo, hi)]
    exp = [_py_quantize(x) for x in w]
    assert got == exp
    # sums of many quantised weights, then one rounding
    sums, acc = [], 0
    for k, q in enumerate(exp):
        acc += q
        if k % 7 == 0:
            sums.append(acc)
    sums += [1, -1, (1 << 24) + 1, (1 << 25) + 2, (1 << 25) + 6, (1 << 100) + (1 << 76), -(1 << 90) - 1]
    lo = np.array([s & ((1 << 64) - 1) for s in sums], np.uint64)
    hi = np.array([s >> 64 for s in sums], np.int64)
    got = oracle.acc_to_float(lo, hi)
    exp_f = np.array([_py_to_f32(s) for s in sums], np.float32)
    np.testing.assert_array_equal(got, exp_f)


def test_pcg32_known_answer(oracle):
    # O'Neill's pcg32 demo: srandom(42, 54) -> 0xa15c02b7 0x7b47f409 0xba1d3330 0x83d2f293 0xbfa4784b 0xcbed606e
    MUL = 0x5851F42D4C957F2D
    M64 = (1 << 64) - 1
    inc = ((54 << 1) | 1) & M64
    state = 0
    state = (state * MUL + inc) & M64
    state = (state + 42) & M64
    state = (state * MUL + inc) & M64
    st = np.array([state], np.uint64)
    ic = np.array([inc], np.uint64)
    expect = [0xA15C02B7, 0x7B47F409, 0xBA1D3330, 0x83D2F293, 0xBFA4784B, 0xCBED606E]
    for e in expect:
        f = oracle.rng_next_f32(st, ic)[0]
        # next_float32 = asfloat((u >> 9) | 0x3f800000) - 1
        want = np.array([(e >> 9) | 0x3F800000], np.uint32).view(np.float32)[0] - np.float32(1)
        assert f == want and 0.0 <= f < 1.0


def test_rng_streams_are_distinct_and_reproducible(oracle):
    s1, i1 = oracle.rng_seed(1000, 7)
    s2, i2 = oracle.rng_seed(1000, 7)
    np.testing.assert_array_equal(s1, s2)
    np.testing.assert_array_equal(i1, i2)
    assert len(set(zip(s1.tolist(), i1.tolist()))) == 1000
    assert (i1 & np.uint64(1)).all()
    s3, _ = oracle.rng_seed(1000, 8)
    assert (s1 != s3).any()
    # lane0 offset continues the lane numbering
    s4, i4 = oracle.rng_seed(10, 7, lane0=990)
    np.testing.assert_array_equal(s4, s1[990:])
