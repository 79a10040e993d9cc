"""Scenes drawn at random (tests/mesh_scenes.py:random_scene): the restated algorithm, the unmodified reference and the
product's device functions must agree on whatever geometry, material mix and light set comes up -- not only on the
Cornell boxes the reference ships (src/scene.hxx:244-385).

not gpu: oracle == unmodified reference (tape replay) == the device functions compiled for the host, bit for bit.
gpu:     the device == oracle (tape, counters, framebuffer), on more seeds and two algorithms each."""
import os

import numpy as np
import pytest

import oracle_lib
from oracle_lib import Oracle
from emul_lib import Emul
from mesh_scenes import random_scene

needs_ref = pytest.mark.skipif(not oracle_lib.have_ref(), reason="oracle/_ref not built (no /root/reference)")
COUNTERS = ("lightVertices", "lightRays", "cameraRays", "shadowRays", "mergeQueries", "mergeCandidates", "mergeAccepted",
            "connections", "lightSplats")
ALGOS = (4, 2, 3, 5, 1, 0, 6)   # VCM, BPM, BPT, PathTracer, PPM, light tracing, EyeLight (include/smallvcm_amd.h vcm_algorithm)


def _oracle_run(sc, algo, nit):
    o = Oracle(sc, algo, threads=min(8, os.cpu_count() or 1))
    lcs, ccs = [], []
    for it in range(nit):
        o.run_iteration(it, 0, 10)
        a, b = o.counts()
        lcs.append(a)
        ccs.append(b)
    return o, lcs, ccs


@needs_ref
@pytest.mark.parametrize("seed", range(14))
def test_oracle_equals_the_unmodified_reference_on_random_scenes(seed):
    algo = ALGOS[seed % len(ALGOS)]
    sc = random_scene(seed, 40, 40)
    assert oracle_lib.ref_check_scene2(sc) == 0
    o, lcs, ccs = _oracle_run(sc, algo, 2)
    fb, consumed, bad = oracle_lib.ref_run_tape2(sc, algo, np.concatenate(lcs), np.concatenate(ccs), n_iter=2)
    assert bad == 0
    assert np.array_equal(fb.view(np.uint32), o.framebuffer().view(np.uint32))
    assert np.isfinite(fb).all()


@pytest.mark.parametrize("seed", range(14))
def test_device_functions_equal_the_oracle_on_random_scenes(seed):
    algo = ALGOS[seed % len(ALGOS)]
    sc = random_scene(seed, 40, 40)
    o, e = Oracle(sc, algo, threads=min(8, os.cpu_count() or 1)), Emul(sc, algo)
    for it in range(2):
        o.run_iteration(it, 0, 10)
        e.run_iteration(it, 0, 10)
    assert np.array_equal(o.framebuffer().view(np.uint32), e.framebuffer().view(np.uint32))
    for a, b in zip(o.counts(), e.counts()):
        assert np.array_equal(a, b)
    so, se = o.stats(), e.stats()
    for k in se:
        assert so[k] == se[k], k


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(24))
def test_gpu_equals_the_oracle_on_random_scenes(seed):
    from smallvcm_amd.renderer import VertexCM
    res = 96 if seed % 3 else 160
    sc = random_scene(seed, res, res)
    for algo in (4, ALGOS[1 + seed % 6]):
        o = Oracle(sc, algo, threads=os.cpu_count() or 1)
        r = VertexCM(sc, algo, 0.003, 0.75, 1234)
        r.mMaxPathLength = 10
        for it in range(2):
            o.run_iteration(it, 0, 10)
            r.RunIteration(it)
            lc, cc = r.backend.rng_counts()
            olc, occ = o.counts()
            assert np.array_equal(lc, olc) and np.array_equal(cc, occ), (seed, algo, it)
            so, sg = o.stats(), r.stats()
            for k in COUNTERS:
                assert so[k] == sg[k], (seed, algo, it, k, so[k], sg[k])
        fb = r.framebuffer_sum()
        r.close()
        assert np.array_equal(fb.view(np.uint32), o.framebuffer().view(np.uint32)), (seed, algo)
