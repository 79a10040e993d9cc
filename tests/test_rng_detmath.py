"""Numeric spec shared by the product and the checker: Philox4x32-10 stream and
the deterministic sinf/cosf/powf.  CPU only."""
import ctypes as C

import numpy as np

from emul_lib import emul
from oracle_lib import oracle


def _philox(ctr, key):
    L = oracle()
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    L.oracle_philox(c, k, o)
    return list(o)


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    assert _philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert _philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert _philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_path_stream_product_equals_oracle():
    L, E = oracle(), emul()
    for seed, it, path, kind in [(1234, 0, 0, 0), (1234, 3, 77, 1), (1, 0, 4194303, 1), (99, 7, 123456, 0)]:
        for k in range(0, 40):
            a = L.oracle_path_float(seed, it, path, kind, k)
            b = E.emul_path_float(seed, it, path, kind, k)
            assert a == b and 0.0 < a < 1.0


def test_stream_is_uniform():
    L = oracle()
    v = np.array([L.oracle_path_float(1234, 0, p, 0, k) for p in range(2000) for k in range(8)])
    assert abs(v.mean() - 0.5) < 0.01 and abs(v.var() - 1 / 12.0) < 0.005


def _ulp_diff(a, b):
    a = np.asarray(a, np.float32).view(np.int32).astype(np.int64)
    b = np.asarray(b, np.float32).view(np.int32).astype(np.int64)
    return np.abs(a - b)


def _ulp_err(approx, exact):
    """error of the binary32 values `approx` in units of the last place of the exact (binary64) result"""
    exact = np.asarray(exact, np.float64)
    return np.abs(np.asarray(approx, np.float64) - exact) / np.spacing(np.abs(exact).astype(np.float32)).astype(np.float64)


def _vec(fn, *cols):
    return np.array([fn(*[float(v) for v in row]) for row in zip(*cols)], np.float32)


def test_detmath_accuracy_and_product_equality():
    """The definition's stated bounds (detmath.h): sinf/cosf <= 1.6 ulp on the domain the path uses, powf <= 1.9 ulp for
    0 < y < 1 and correctly rounded for integer exponents; product (host build of smallvcm_amd/csrc/detmath.h) ==
    oracle (oracle/detmath_ref.h) bit for bit."""
    L, E = oracle(), emul()
    rng = np.random.default_rng(7)
    xs = np.concatenate([(rng.random(40000) * 2 * np.pi).astype(np.float32),
                         (rng.random(8000) * 9 - 1.5).astype(np.float32),
                         np.array([0.0, 1e-8, np.pi / 4, np.pi / 2, np.pi, 1.5 * np.pi, 2 * np.pi, 6.2831855, -0.7853982], np.float32)])
    s, c = _vec(L.oracle_sinf, xs), _vec(L.oracle_cosf, xs)
    x64 = xs.astype(np.float64)
    assert _ulp_err(s, np.sin(x64)).max() <= 1.6 and _ulp_err(c, np.cos(x64)).max() <= 1.6
    assert np.abs(s.astype(np.float64) - np.sin(x64)).max() < 1.2e-7 and np.abs(c.astype(np.float64) - np.cos(x64)).max() < 1.2e-7
    assert (_ulp_err(s, np.sin(x64)) > 1).mean() < 0.02       # a percent of the arguments is off by more than one ulp
    assert np.array_equal(_vec(E.emul_sinf, xs[:6000]), s[:6000]) and np.array_equal(_vec(E.emul_cosf, xs[:6000]), c[:6000])
    assert np.all(np.abs(s.astype(np.float64) ** 2 + c.astype(np.float64) ** 2 - 1) < 4e-7)
    # powf, fractional exponents: the stream's floats (2k+1) 2^-24, anything in (0, 200), and the whole binary32 range
    us = ((2 * rng.integers(0, 1 << 23, 30000) + 1).astype(np.float64) * 2.0 ** -24).astype(np.float32)
    wide = np.exp(rng.random(30000) * 170 - 85).astype(np.float32)
    for y in (float(np.float32(1.0 / 91.0)), 0.0625, 0.125, float(np.float32(1 / 2.2)), 0.5, 0.99):
        for x in (us, wide):
            p = _vec(lambda a: L.oracle_powf(a, y), x)
            assert _ulp_err(p, np.power(x.astype(np.float64), float(np.float32(y)))).max() <= 1.9, y
    ys = rng.random(30000).astype(np.float32)
    p = _vec(L.oracle_powf, wide, ys)
    assert _ulp_err(p, np.power(wide.astype(np.float64), ys.astype(np.float64))).max() <= 1.9
    assert np.array_equal(_vec(E.emul_powf, wide[:6000], ys[:6000]), p[:6000])
    # integer exponents: one rounding of the binary64 product chain, i.e. the correctly rounded result
    u01 = np.concatenate([rng.random(20000).astype(np.float32), np.array([0.0, 1.0, 1e-3, 1.0000001, 1e-30], np.float32)])
    for y in (90.0, 1.0, 2.0, 3.0, 17.0, 256.0, 1000.0):
        p = _vec(lambda a: L.oracle_powf(a, y), u01)
        assert _ulp_err(p[p > 1e-37], np.power(u01[p > 1e-37].astype(np.float64), y)).max() <= 0.5001, y
        assert np.array_equal(_vec(lambda a: E.emul_powf(a, y), u01[:3000]), p[:3000])
    # mixed and negative exponents, special cases
    for y in (90.5, 2.25, -0.5, -3.0):
        x = (rng.random(5000) + 0.6).astype(np.float32) if y > 10 else (rng.random(5000) * 3 + 0.01).astype(np.float32)
        p = _vec(lambda a: L.oracle_powf(a, y), x)
        assert _ulp_err(p, np.power(x.astype(np.float64), y)).max() <= 3.0, y
        assert np.array_equal(_vec(lambda a: E.emul_powf(a, y), x[:2000]), p[:2000])
    assert L.oracle_powf(0.0, 90.0) == 0.0 and L.oracle_powf(5.0, 0.0) == 1.0 and L.oracle_powf(1.0, 3.3) == 1.0
    assert L.oracle_powf(-2.0, 0.5) == 0.0 and L.oracle_powf(1e-45, 0.5) > 0.0
    # the radius schedule (vertexcm.hxx:296): (i + 1)^0.125
    it = np.arange(1, 20001).astype(np.float32)
    assert _ulp_err(_vec(lambda a: L.oracle_powf(a, 0.125), it), np.power(it.astype(np.float64), 0.125)).max() <= 1.9
