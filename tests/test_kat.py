"""T0 -- function-level known answers: single calls of the product's device functions
(smallvcm_amd/csrc/vcm_kat.h) against the REFERENCE's own classes (oracle/_ref: BSDF<>, the four lights,
Scene::Intersect / Occluded, Camera), bit for bit, on seeded inputs.  A parity break in an iteration test is
localised here: the op that differs names the function.

  not gpu:  device functions compiled for the host (tests/host_emul)  vs  reference
  gpu:      the same functions ON THE DEVICE (vcm_debug_kat)          vs  reference (or vs the host build when
            oracle/_ref is not shipped)
"""
import ctypes as C

import numpy as np
import pytest

import oracle_lib
from emul_lib import emul
from smallvcm_amd.renderer import cornell_scene

KAT_FLOATS = 16
OPS = {"intersect": 0, "occluded": 1, "bsdf_eval": 2, "bsdf_sample": 3, "light_emit": 4, "light_illuminate": 5,
       "light_radiance": 6, "camera": 7}
# scene 0..3 of g_SceneConfigs + every light type at once with the large glass sphere
MASKS = {"s0-sun": 2 | 64 | 128 | 256, "s1-ceiling": 1 | 16 | 256, "s2-point": 4 | 64 | 128 | 256, "s3-background": 8 | 64 | 128 | 256,
         "all-lights-glass": 1 | 2 | 4 | 8 | 32}
_fp = C.POINTER(C.c_float)


def _unit(rng, n):
    v = rng.normal(size=(n, 3)).astype(np.float32)
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)


def _inside(rng, n):
    """points inside the Cornell box (corners +-1.27 etc., scene.hxx:196-205)"""
    return (rng.random((n, 3)) * np.float32([2.4, 2.4, 2.3]) - np.float32([1.2, 1.2, 1.15])).astype(np.float32)


def make_inputs(op, scene, n, seed):
    rng = np.random.default_rng(seed)
    a = np.zeros((n, KAT_FLOATS), np.float32)
    nmat, nlights = scene.nMaterials, scene.nLights
    if op in ("intersect", "occluded"):
        a[:, 0:3] = _inside(rng, n)
        a[:, 3:6] = _unit(rng, n)
        a[:, 6] = 0.0 if op == "intersect" else (rng.random(n) * 3).astype(np.float32)
        # a share of axis-aligned / edge-grazing rays: the tie rules of geometry.hxx:65-91, :133-142
        k = n // 8
        a[:k, 3:6] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, k)] * rng.choice(np.float32([-1, 1]), (k, 1))
        a[k:2 * k, 0:3] = np.round(a[k:2 * k, 0:3] * 4) / 4
    elif op in ("bsdf_eval", "bsdf_sample"):
        a[:, 0:3] = _unit(rng, n)
        a[:, 3:6] = _unit(rng, n)
        k = n // 4   # axis-aligned normals as the box has them, incl. the |z.x| > 0.99 frame switch (frame.hxx:56)
        a[:k, 3:6] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, k)] * rng.choice(np.float32([-1, 1]), (k, 1))
        a[:, 6] = rng.integers(0, nmat, n)
        # half of the records name a triangle of the scene (normal, material and index of that primitive): the product
        # then takes frame and component probabilities from its per-scene tables instead of computing them
        tris = [i for i in range(scene.nPrims) if scene.prims[i].type == 0]
        pick = rng.choice(tris, n // 2)
        a[n // 2:n // 2 + len(pick), 3:6] = np.float32([list(scene.prims[i].n) for i in pick])
        a[n // 2:n // 2 + len(pick), 6] = [scene.prims[i].matID for i in pick]
        a[n // 2:n // 2 + len(pick), 11] = pick + 1
        if op == "bsdf_eval":
            a[:, 7:10] = _unit(rng, n)
            m = n // 4   # directions near the mirror direction: the Phong lobe (exponent 90) is non-zero there
            refl = a[k:k + m, 0:3] - 2 * np.sum(a[k:k + m, 0:3] * a[k:k + m, 3:6], axis=1, keepdims=True) * a[k:k + m, 3:6]
            refl = refl + 0.05 * rng.normal(size=refl.shape)
            a[k:k + m, 7:10] = (refl / np.linalg.norm(refl, axis=1, keepdims=True)).astype(np.float32)
        else:
            a[:, 7:10] = rng.random((n, 3))
            a[:, 10] = rng.integers(0, 2, n)
    elif op == "light_emit":
        a[:, 0] = rng.integers(0, nlights + 1, n)   # nlights itself: GetLightPtr clamps (scene.hxx:98-102)
        a[:, 1:5] = rng.random((n, 4))
    elif op == "light_illuminate":
        a[:, 0] = rng.integers(0, nlights + 1, n)
        a[:, 1:4] = _inside(rng, n)
        a[:, 4:6] = rng.random((n, 2))
    elif op == "light_radiance":
        a[:, 0] = rng.integers(0, nlights, n)
        a[:, 1:4] = _unit(rng, n)
        a[:, 4:7] = _inside(rng, n)
    elif op == "camera":
        res = scene.camera.resolution
        a[:, 0] = rng.random(n) * res[0]
        a[:, 1] = rng.random(n) * res[1]
        a[:, 2:5] = _inside(rng, n) * 1.2
    return np.ascontiguousarray(a)


def ref_answers(mask, res, op, inp):
    out = np.zeros_like(inp)
    L = oracle_lib.ref_tape()
    L.ref_kat.argtypes = [C.c_uint, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp]
    assert L.ref_kat(mask, res, res, OPS[op], len(inp), inp.ctypes.data_as(_fp), out.ctypes.data_as(_fp)) == 0
    return out


def host_answers(scene, op, inp):
    out = np.zeros_like(inp)
    E = emul()
    E.emul_kat.argtypes = [C.c_void_p, C.c_int, C.c_int, _fp, _fp]
    E.emul_kat(C.byref(scene), OPS[op], len(inp), inp.ctypes.data_as(_fp), out.ctypes.data_as(_fp))
    return out


def _diff(op, a, b):
    bad = np.nonzero((a.view(np.uint32) != b.view(np.uint32)).any(axis=1))[0]
    return "%s: %d of %d records differ, first %s" % (op, len(bad), len(a), bad[:5])


@pytest.mark.skipif(not oracle_lib.have_ref(), reason="oracle/_ref not built (no /root/reference)")
@pytest.mark.parametrize("name", sorted(MASKS))
@pytest.mark.parametrize("op", sorted(OPS))
def test_device_functions_on_the_host_equal_the_reference_classes(op, name):
    mask, res = MASKS[name], 64
    scene = cornell_scene(mask, res, res, is_mask=True)
    inp = make_inputs(op, scene, 20000, seed=OPS[op] * 31 + len(name))
    ref, mine = ref_answers(mask, res, op, inp), host_answers(scene, op, inp)
    assert np.array_equal(ref.view(np.uint32), mine.view(np.uint32)), _diff(op, ref, mine)
    if not (op == "light_radiance" and name in ("s0-sun", "s2-point")):   # delta lights are never hit: all-zero answers
        assert np.count_nonzero(ref) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(MASKS))
@pytest.mark.parametrize("op", sorted(OPS))
def test_device_functions_on_the_device_equal_the_reference_classes(op, name):
    from smallvcm_amd.renderer import HipBackend
    mask, res = MASKS[name], 64
    scene = cornell_scene(mask, res, res, is_mask=True)
    inp = make_inputs(op, scene, 50000, seed=1000 + OPS[op] * 31 + len(name))
    b = HipBackend(scene, 4, 0.003, 0.75, 1234)
    b.L.vcm_debug_kat.argtypes = [C.c_void_p, C.c_int, C.c_int, _fp, _fp]
    dev = np.zeros_like(inp)
    assert b.L.vcm_debug_kat(b.ctx, OPS[op], len(inp), inp.ctypes.data_as(_fp), dev.ctypes.data_as(_fp)) == 0, b.L.vcm_last_error()
    b.close()
    want = ref_answers(mask, res, op, inp) if oracle_lib.have_ref() else host_answers(scene, op, inp)
    assert np.array_equal(want.view(np.uint32), dev.view(np.uint32)), _diff(op, want, dev)
