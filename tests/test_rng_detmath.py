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


def test_detmath_accuracy_and_product_equality():
    """<= 1 ulp from the correctly rounded value on the domains the path uses;
    product (host build of smallvcm_amd/csrc/detmath.h) == oracle bit for bit."""
    L, E = oracle(), emul()
    rng = np.random.default_rng(7)
    xs = np.concatenate([(rng.random(20000) * 2 * np.pi).astype(np.float32),
                         (rng.random(2000) * 9 - 1.5).astype(np.float32),
                         np.array([0.0, 1e-8, np.pi / 2, np.pi, 1.5 * np.pi, 2 * np.pi, 6.2831855], np.float32)])
    s = np.array([L.oracle_sinf(float(x)) for x in xs], np.float32)
    c = np.array([L.oracle_cosf(float(x)) for x in xs], np.float32)
    assert _ulp_diff(s, np.sin(xs.astype(np.float64)).astype(np.float32)).max() <= 1
    assert _ulp_diff(c, np.cos(xs.astype(np.float64)).astype(np.float32)).max() <= 1
    assert all(E.emul_sinf(float(x)) == L.oracle_sinf(float(x)) for x in xs[:4000])
    assert all(E.emul_cosf(float(x)) == L.oracle_cosf(float(x)) for x in xs[:4000])
    us = np.concatenate([rng.random(20000).astype(np.float32), np.array([0.0, 1.0, 1e-3, 1.0000001, 1e-30], np.float32)])
    for y in (90.0, float(np.float32(1.0 / 91.0)), 1.0, 0.0625, 2.0):
        p = np.array([L.oracle_powf(float(u), y) for u in us], np.float32)
        ref = np.power(us.astype(np.float64), y).astype(np.float32)
        assert _ulp_diff(p, ref).max() <= 1, y
        assert all(E.emul_powf(float(u), y) == L.oracle_powf(float(u), y) for u in us[:3000])
    assert L.oracle_powf(0.0, 90.0) == 0.0 and L.oracle_powf(5.0, 0.0) == 1.0 and L.oracle_powf(1.0, 3.3) == 1.0
