"""Pins the oracle (oracle/vcm_oracle.cpp): bit-exact against
  (a) committed golden framebuffers rendered by the UNMODIFIED reference with
      the tape-replay Rng (tests/golden/fb_goldens.npz, gen_golden.py), and
  (b) the live reference build oracle/_ref when it is present."""
import os
import re

import numpy as np
import pytest

import oracle_lib
from oracle_lib import Oracle, ref_run_tape, ref_scene
from smallvcm_amd._abi import SCENE_CONFIGS, SceneDesc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_G = np.load(os.path.join(GOLD, "fb_goldens.npz"))
_KEYS = sorted(k[:-3] for k in _G.files if k.endswith("_fb"))


def _parse(key):
    m = re.match(r"s(\d+)_a(\d+)_r(\d+)_i(\d+)_l(\d+)_(\d+)", key)
    return tuple(int(x) for x in m.groups())


def golden_scene(sid, res):
    return SceneDesc.frombytes(open(os.path.join(GOLD, "scene_%d_%d.bin" % (sid, res)), "rb").read())


@pytest.mark.parametrize("key", _KEYS)
def test_oracle_matches_reference_golden(key):
    sid, algo, res, nit, mn, mx = _parse(key)
    from smallvcm_amd.renderer import cornell_scene
    sc = cornell_scene(sid, res, res)
    o = Oracle(sc, algo)
    lcs, ccs = [], []
    for it in range(nit):
        o.run_iteration(it, mn, mx)
        a, b = o.counts()
        lcs.append(a)
        ccs.append(b)
    assert np.array_equal(np.concatenate(lcs), _G[key + "_lc"]), "random-number tape differs"
    assert np.array_equal(np.concatenate(ccs), _G[key + "_cc"])
    assert np.array_equal(o.framebuffer(), _G[key + "_fb"])


needs_ref = pytest.mark.skipif(not oracle_lib.have_ref(), reason="oracle/_ref not built (no /root/reference)")


@needs_ref
@pytest.mark.parametrize("sid,algo,res,nit,mn,mx,threads", [
    (1, 4, 128, 2, 0, 10, 4), (3, 4, 96, 2, 0, 10, 1), (0, 4, 96, 1, 0, 10, 4), (2, 4, 96, 1, 0, 10, 4),
    (1, 2, 96, 1, 0, 10, 4), (1, 3, 64, 2, 1, 7, 1), (2, 1, 64, 2, 0, 10, 4), (0, 0, 64, 1, 0, 10, 1),
    # PathTracer (pathtracer.hxx) and EyeLight (eyelight.hxx; iteration 1 draws no jitter) behind the same interface
    (0, 5, 96, 2, 0, 10, 4), (1, 5, 96, 1, 0, 10, 4), (2, 5, 64, 1, 0, 10, 1), (3, 5, 96, 2, 0, 10, 4), (1, 5, 64, 1, 2, 5, 1),
    (0, 6, 64, 3, 0, 10, 4), (3, 6, 64, 2, 0, 10, 1)])
def test_oracle_matches_live_reference(sid, algo, res, nit, mn, mx, threads):
    mask = SCENE_CONFIGS[sid]
    sc = ref_scene(mask, res, res)
    o = Oracle(sc, algo, threads=threads)
    lcs, ccs = [], []
    for it in range(nit):
        o.run_iteration(it, mn, mx)
        a, b = o.counts()
        lcs.append(a)
        ccs.append(b)
    fb, consumed, bad = ref_run_tape(mask, res, res, algo, np.concatenate(lcs), np.concatenate(ccs), n_iter=nit,
                                     min_len=mn, max_len=mx)
    assert bad == 0
    assert consumed == int(np.concatenate(lcs).sum()) + (0 if algo == 0 else int(np.concatenate(ccs).sum()))
    assert np.array_equal(fb, o.framebuffer())


# Box masks OUTSIDE g_SceneConfigs (scene.hxx:112-126): large glass sphere on a diffuse floor; kDefault; every light
# type at once (2 area + sun + point + background); both large spheres requested (the loader keeps the mirror one);
# and long paths (maxPathLength 24: more than 15 stored vertices per light path occur).
MASK_CASES = [(1 | 32, 4, 80, 2, 0, 10), (1 | 64 | 128, 4, 64, 1, 0, 10), (1 | 2 | 4 | 8 | 256 | 128, 4, 64, 2, 0, 10),
              (1 | 2 | 4 | 8 | 256 | 128, 2, 48, 1, 0, 10), (16 | 32 | 1 | 256, 3, 64, 1, 0, 10), (1 | 32, 4, 48, 1, 0, 24),
              (8 | 64, 5, 64, 1, 0, 10)]


@needs_ref
@pytest.mark.parametrize("mask,algo,res,nit,mn,mx", MASK_CASES)
def test_oracle_matches_live_reference_on_other_box_masks(mask, algo, res, nit, mn, mx):
    sc = ref_scene(mask, res, res)
    o = Oracle(sc, algo, threads=4)
    lcs, ccs = [], []
    for it in range(nit):
        o.run_iteration(it, mn, mx)
        a, b = o.counts()
        lcs.append(a)
        ccs.append(b)
    fb, consumed, bad = ref_run_tape(mask, res, res, algo, np.concatenate(lcs), np.concatenate(ccs), n_iter=nit,
                                     min_len=mn, max_len=mx)
    assert bad == 0
    assert np.array_equal(fb, o.framebuffer())
    assert fb.max() > 0


@needs_ref
def test_libm_interposition_is_active():
    before = oracle_lib.ref_tape().ref_detmath_calls()
    sc = ref_scene(SCENE_CONFIGS[1], 16, 16)
    o = Oracle(sc, 4)
    o.run_iteration(0)
    lc, cc = o.counts()
    ref_run_tape(SCENE_CONFIGS[1], 16, 16, 4, lc, cc)
    assert oracle_lib.ref_tape().ref_detmath_calls() > before


@needs_ref
def test_tape_desync_is_detected():
    """A wrong tape (one path claims one float too few) must be flagged."""
    sc = ref_scene(SCENE_CONFIGS[1], 16, 16)
    o = Oracle(sc, 4)
    o.run_iteration(0)
    lc, cc = o.counts()
    lc = lc.copy()
    lc[5] -= 1
    fb, consumed, bad = ref_run_tape(SCENE_CONFIGS[1], 16, 16, 4, lc, cc)
    assert bad == 1 or not np.array_equal(fb, o.framebuffer())


@needs_ref
def test_statistical_agreement_with_stock_reference():
    """T2: the counter-based RNG + deterministic libm do not bias the estimator:
    image means of the oracle agree with the STOCK reference (mt19937, glibc)."""
    res, nit = 48, 24
    for sid, algo in [(1, 4), (3, 4), (1, 2), (2, 3)]:
        mask = SCENE_CONFIGS[sid]
        sc = ref_scene(mask, res, res)
        o = Oracle(sc, algo, threads=4)
        for it in range(nit):
            o.run_iteration(it)
        mine = o.framebuffer() / nit
        ref, _ = oracle_lib.ref_render_stock(mask, res, res, algo, iterations=nit, threads=1)
        m1, m2 = mine.mean(axis=(0, 1)), ref.mean(axis=(0, 1))
        assert np.all(np.abs(m1 - m2) < 0.04 * np.maximum(m2, 0.05)), (sid, algo, m1, m2)


def test_deterministic_libm_against_the_hosts_libm():
    """north_star: "per-pixel RMSE < 1e-4 vs reference at fixed seed" -- against the reference AS SHIPPED, i.e. with its
    own libm.  Since round 4 detmath restates that libm (glibc 2.35; one deviation: integer exponents are the correctly
    rounded power), so the bound is asserted directly, at BASELINE's 512^2 (C1) and on the three other scenes:
      (1) the UNMODIFIED reference, its own libm (oracle/_ref/libsmallvcm_ref_tape_libm.so: nothing interposed), replays
          the oracle's tape: no desynchronisation -- every path draws the same number of random floats -- and one
          iteration's framebuffer within RMSE 1e-6 (measured: 1e-10 .. 2e-9, profiles/archive/r06_libm_tolerance.json);
      (2) the oracle built over the host's libm calls (liboracle_glibc.so) against the checker: the same bound, no path
          with a different float count."""
    import os
    import subprocess
    import numpy as np
    from smallvcm_amd._abi import SCENE_CONFIGS
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-C", os.path.join(root, "oracle"), "glibc"], check=True, capture_output=True)
    glibc = oracle_lib.load_oracle(os.path.join(root, "oracle", "liboracle_glibc.so"))
    for sid, algo, res in ((1, 4, 512), (3, 4, 192), (0, 2, 160), (2, 3, 160), (1, 1, 128)):
        sc = oracle_lib.ref_scene(SCENE_CONFIGS[sid], res, res)
        o = oracle_lib.Oracle(sc, algo, threads=8)
        o.run_iteration(0, 0, 10)
        lc, cc = o.counts()
        fb = o.framebuffer().astype(np.float64)
        ref, consumed, bad = oracle_lib.ref_run_tape(SCENE_CONFIGS[sid], res, res, algo, lc, cc, own_libm=True)
        assert bad == 0 and consumed == int(lc.sum()) + int(cc.sum()), (sid, algo, res)
        rmse = np.sqrt(((fb - ref) ** 2).mean())
        assert rmse < 1e-6, (sid, algo, res, rmse)          # north_star's bound is 1e-4
        assert (np.abs(fb - ref).max(axis=2) > 0).mean() < 0.01    # and almost every pixel is the reference's bit for bit
        g = oracle_lib.Oracle(sc, algo, threads=8, lib=glibc)
        g.run_iteration(0, 0, 10)
        glc, gcc = g.counts()
        assert np.array_equal(glc, lc) and np.array_equal(gcc, cc)
        assert np.sqrt(((g.framebuffer().astype(np.float64) - fb) ** 2).mean()) < 1e-6
        # the interposed build (what every bit-exact test replays into) is the checker bit for bit
        ref2, _, bad2 = oracle_lib.ref_run_tape(SCENE_CONFIGS[sid], res, res, algo, lc, cc)
        assert bad2 == 0 and np.array_equal(ref2, o.framebuffer())
