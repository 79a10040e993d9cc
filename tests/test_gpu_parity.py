"""GPU parity: the HIP path (through the C-ABI) against the oracle and against
the unmodified reference (oracle/_ref, replaying the GPU's random-number tape).

Bars
  * default (wavefront) mode: framebuffer BIT-EXACT for all five algorithms
    (BASELINE.json asks for RMSE < 1e-4);
  * strict mode: bit-exact for bpm / ppm; lt / bpt / vcm splat with fp32 atomic
    adds whose order is not defined: |d| <= 2e-5*|v| + 2e-7, RMSE < 1e-6;
  * random-number tape, merge records, hash grid (cell ranges AND in-cell
    order) and workload counters: bit-exact / equal.
"""
import numpy as np
import pytest

import oracle_lib
from oracle_lib import Oracle
from smallvcm_amd._abi import SCENE_CONFIGS
from smallvcm_amd.renderer import VertexCM, cornell_scene, load_library

pytestmark = pytest.mark.gpu

SPLAT_ALGOS = (0, 3, 4)


def _rmse(a, b):
    return float(np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)))


def _check_fb(gpu, ref, algo, strict=True):
    """Default (wavefront) mode: BIT-EXACT for every algorithm -- connections, merges and
    light splats are all added in the reference's order.  Strict mode splats with fp32
    atomics (order not defined): rounding-level differences for lt / bpt / vcm."""
    if not strict or algo not in SPLAT_ALGOS:
        assert np.array_equal(gpu, ref)
    else:
        assert np.all(np.abs(gpu - ref) <= 2e-5 * np.abs(ref) + 2e-7), float(np.abs(gpu - ref).max())
        assert _rmse(gpu, ref) < 1e-6


def test_numeric_spec_on_device():
    """IEEE divide/sqrt, no FMA contraction, detmath and Philox: device == host checker, bit for bit."""
    import ctypes as C
    L = load_library()
    O = oracle_lib.oracle()
    rng = np.random.default_rng(3)
    n = 1 << 16
    fp = C.POINTER(C.c_float)

    def dev(op, a, b):
        out = np.zeros(n, np.float32)
        assert L.vcm_debug_numeric_spec(op, n, a.ctypes.data_as(fp), b.ctypes.data_as(fp), out.ctypes.data_as(fp)) == 0
        return out
    a = (rng.random(n) * 7 - 0.5).astype(np.float32)
    b = (rng.random(n) * 100 + 1e-3).astype(np.float32)

    def same(x, y):   # bit for bit, any NaN = any NaN
        x, y = np.asarray(x, np.float32), np.asarray(y, np.float32)
        nan = np.isnan(x)
        return np.array_equal(nan, np.isnan(y)) and np.array_equal(x[~nan].view(np.uint32), y[~nan].view(np.uint32))
    M = C.CDLL("libm.so.6")   # the libm of the reference's image: detmath restates it (tests/test_rng_detmath.py)
    M.sinf.restype = M.cosf.restype = M.powf.restype = C.c_float
    M.sinf.argtypes = M.cosf.argtypes = [C.c_float]
    M.powf.argtypes = [C.c_float, C.c_float]
    # sinf / cosf: the path's range, the large reduction, every kind of bit pattern
    wide = np.concatenate([a[: n // 4], (rng.standard_normal(n // 4) * 60).astype(np.float32),
                           np.exp(rng.random(n // 4) * 80 - 40).astype(np.float32),
                           rng.integers(0, 1 << 32, n // 4, dtype=np.uint64).astype(np.uint32).view(np.float32)])
    wide[:12] = np.array([0.0, -0.0, 2.0 ** -12, 1e-30, 0.75, np.pi / 4, 119.99999, 120.0, 3e38, np.inf, -np.inf, np.nan], np.float32)
    for op, of, mf in ((0, O.oracle_sinf, M.sinf), (1, O.oracle_cosf, M.cosf)):
        d = dev(op, wide, b)
        assert same(d, np.array([of(float(x)) for x in wide], np.float32))
        if oracle_lib.host_libm_is_the_restated_one():   # the ambient libm is glibc 2.35 / FMA: the one detmath restates
            assert same(d, np.array([mf(float(x)) for x in wide], np.float32))
    u = rng.random(n).astype(np.float32)
    y = np.where(rng.random(n) < 0.5, 90.0, 1.0 / 91.0).astype(np.float32)
    assert np.array_equal(dev(2, u, y), np.array([O.oracle_powf(float(p), float(q)) for p, q in zip(u, y)], np.float32))
    # powf, the general path (table walk, subnormals, overflow / underflow, special cases) against oracle and host libm
    px = np.concatenate([np.exp(rng.random(n // 2) * 170 - 85).astype(np.float32),
                         rng.integers(0, 1 << 32, n // 2, dtype=np.uint64).astype(np.uint32).view(np.float32)])
    py = np.concatenate([(rng.random(n // 2) * 8 - 4).astype(np.float32),
                         rng.integers(0, 1 << 32, n // 2, dtype=np.uint64).astype(np.uint32).view(np.float32)])
    sp = [(0.0, 0.5), (-0.0, 0.5), (0.0, -0.5), (-2.0, 0.5), (-2.0, 3.0), (1e-45, 0.5), (1.0, 3.3), (5.0, 0.0), (np.inf, 0.5), (np.inf, -0.5),
          (0.5, np.inf), (2.0, np.inf), (2.0, -np.inf), (np.nan, 0.5), (2.0, np.nan), (3e38, 1.5), (1e-30, 1.5), (0.5, 149.5), (0.5, 150.5),
          (0.5, 140.25), (1e-40, 0.3), (0.999, 90.5), (0.3, 70000.0)]
    px[: len(sp)] = np.array([q[0] for q in sp], np.float32)
    py[: len(sp)] = np.array([q[1] for q in sp], np.float32)
    d = dev(2, px, py)
    assert same(d, np.array([O.oracle_powf(float(p), float(q)) for p, q in zip(px, py)], np.float32))
    with np.errstate(invalid="ignore"):
        general = ~((py >= 1) & (py <= 65536) & (py == np.floor(py)))   # integer exponents are the correctly rounded power instead
    assert same(d[general], np.array([M.powf(float(p), float(q)) for p, q in zip(px[general], py[general])], np.float32))
    assert same(dev(6, px, py), d)                                   # the same with the tables in LDS
    yi = rng.integers(1, 300, n).astype(np.float32)                  # lanes with different integer exponents
    yi[: n // 2] = 90.0                                              # ... and waves whose exponent is uniform
    assert same(dev(7, u, yi), np.array([O.oracle_powf(float(p), float(q)) for p, q in zip(u, yi)], np.float32))
    fin = np.isfinite(wide)
    assert same(dev(8, wide, b)[fin], np.array([O.oracle_sinf(float(x)) for x in wide], np.float32)[fin])
    assert np.array_equal(dev(3, a, b), a / b)                       # correctly rounded fp32 division
    assert np.array_equal(dev(4, np.abs(a), b), np.sqrt(np.abs(a)))   # correctly rounded fp32 sqrt
    assert np.array_equal(dev(5, a, b), a * b + a)                   # mul then add, no FMA
    out = np.zeros((257, 24), np.float32)
    assert L.vcm_debug_philox_spec(1234, 5, 1, 257, 24, out.ctypes.data_as(fp)) == 0
    for p in (0, 1, 100, 256):
        for k in (0, 3, 4, 23):
            assert out[p, k] == O.oracle_path_float(1234, 5, p, 1, k)


CASES = [(sid, algo, 64, 1, 0, 10) for sid in range(4) for algo in range(5)] + [
    (1, 4, 256, 2, 0, 10), (3, 4, 192, 2, 0, 10), (1, 2, 256, 1, 0, 10), (0, 4, 128, 2, 2, 6), (1, 4, 64, 1, 0, 1),
    (1, 4, 64, 1, 0, 2), (2, 3, 100, 1, 0, 10), (1, 4, 8, 1, 0, 10), (1, 2, 130, 3, 0, 5)]


@pytest.mark.parametrize("strict", [False, True], ids=["wavefront", "strict"])
@pytest.mark.parametrize("sid,algo,res,nit,mn,mx", CASES)
def test_hip_equals_oracle(sid, algo, res, nit, mn, mx, strict):
    sc = cornell_scene(sid, res, res)
    o = Oracle(sc, algo, threads=8)
    r = VertexCM(sc, algo, 0.003, 0.75, 1234, strict_order=strict)
    r.mMinPathLength, r.mMaxPathLength = mn, mx
    for it in range(nit):
        o.run_iteration(it, mn, mx)
        r.RunIteration(it)
        lc, cc = r.backend.rng_counts()
        olc, occ = o.counts()
        assert np.array_equal(lc, olc), "light tape"
        if algo != 0:
            assert np.array_equal(cc, occ), "camera tape"
        so, sg = o.stats(), r.stats()
        for k in ("lightVertices", "lightRays", "cameraRays", "shadowRays", "mergeQueries", "mergeCandidates",
                  "mergeAccepted", "connections", "lightSplats"):
            assert so[k] == sg[k], (k, so[k], sg[k])
        if algo in (1, 2, 4):
            assert np.array_equal(o.records().view(np.uint32), r.backend.records().view(np.uint32)), "merge records"
            ce, idx, bbox = o.grid()
            cs, sidx, gb = r.backend.grid()
            assert np.array_equal(gb, bbox)
            assert cs[0] == 0 and np.array_equal(cs[1:], ce), "cell ranges"
            assert np.array_equal(sidx, idx), "in-cell order (stable counting sort)"
    _check_fb(r.framebuffer_sum(), o.framebuffer(), algo, strict)
    r.close()


def test_long_splat_lists_one_wave_per_pixel():
    """Pixels that receive many light splats in one iteration (a caustic) are ordered by k_splat_apply_long, one wave per
    pixel; with SMALLVCM_AMD_SPLAT_LONG=8 every pixel with more than 8 splats goes that way, at sizes the oracle checks
    in seconds: light tracing and VCM on the point-light / glass-sphere scene at 48^2 (many splats per pixel) and on
    scene 1, framebuffer bit for bit."""
    import os
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import numpy as np, sys; sys.path.insert(0, %r)\n"
            "from smallvcm_amd.renderer import VertexCM, cornell_scene\n"
            "out = []\n"
            "for sid, algo, res, nit in ((2, 0, 48, 3), (2, 4, 48, 2), (1, 4, 96, 2), (0, 3, 64, 2)):\n"
            "    r = VertexCM(cornell_scene(sid, res, res), algo, 0.003, 0.75, 1234); r.mMaxPathLength = 10\n"
            "    for it in range(nit): r.RunIteration(it)\n"
            "    out.append(r.framebuffer_sum().ravel()); r.close()\n"
            "np.save(sys.argv[1], np.concatenate(out))\n") % root
    path = os.path.join(tempfile.mkdtemp(), "fb.npy")
    r = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, SMALLVCM_AMD_SPLAT_LONG="8"), capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    got = np.load(path)
    want = []
    for sid, algo, res, nit in ((2, 0, 48, 3), (2, 4, 48, 2), (1, 4, 96, 2), (0, 3, 64, 2)):
        o = Oracle(cornell_scene(sid, res, res), algo, threads=8)
        for it in range(nit):
            o.run_iteration(it, 0, 10)
        want.append(o.framebuffer().astype(np.float32).ravel())
    want = np.concatenate(want)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert got.max() > 0


@pytest.mark.parametrize("kind", ["walk", "pairs"])
@pytest.mark.parametrize("sid,algo,res,nit,mx", [(1, 4, 256, 3, 10), (3, 2, 192, 2, 10), (0, 1, 128, 2, 6), (2, 4, 64, 1, 10), (1, 4, 40, 1, 10)])
def test_merge_kernels_equal_oracle(sid, algo, res, nit, mx, kind):
    """The two range-merge kernels (vcm_set_merge_kernel) evaluate the accepted photons differently and give the same bits."""
    sc = cornell_scene(sid, res, res)
    o = Oracle(sc, algo, threads=8)
    r = VertexCM(sc, algo, 0.003, 0.75, 1234)
    r.backend.set_merge_kernel(kind)
    r.mMaxPathLength = mx
    for it in range(nit):
        o.run_iteration(it, 0, mx)
        r.RunIteration(it)
        so, sg = o.stats(), r.stats()
        for k in ("mergeQueries", "mergeCandidates", "mergeAccepted"):
            assert so[k] == sg[k], (k, so[k], sg[k])
    assert np.array_equal(r.framebuffer_sum().view(np.uint32), o.framebuffer().astype(np.float32).view(np.uint32))
    r.close()


def test_merge_kernels_agree_on_cells_with_more_than_65535_photons():
    """k_merge_pairs keeps a lane's cell runs in LDS as {start, 16-bit length}; a run of 65535 or more saturates and the lane
    re-reads the cell's end when it gets there (merge_pairs_run_end) -- caustics and point lights next to a wall make such cells.
    A merge radius of 0.6 scene radii puts the whole Cornell box into a handful of cells: 310 000 light vertices, the
    fullest cell far beyond 65535.  Both kernels, same bits, same counters (k_merge_walk keeps 32-bit ranges)."""
    frames, stats, fullest = {}, {}, 0
    for kind in ("walk", "pairs"):
        r = VertexCM(cornell_scene(1, 384, 384), VertexCM.kBpm, 0.6, 0.75, 1234)
        r.backend.set_merge_kernel(kind)
        r.mMaxPathLength = 10
        r.RunIteration(0)
        frames[kind] = r.framebuffer_sum()
        stats[kind] = r.stats()
        cs, _, _ = r.backend.grid()
        fullest = int(np.diff(cs.astype(np.int64)).max())
        r.close()
    assert fullest > 65535 * 1.5, fullest
    for k in ("mergeQueries", "mergeCandidates", "mergeAccepted"):
        assert stats["walk"][k] == stats["pairs"][k], (k, stats["walk"][k], stats["pairs"][k])
    assert stats["pairs"]["mergeAccepted"] > 10_000_000
    assert np.array_equal(frames["walk"].view(np.uint32), frames["pairs"].view(np.uint32))


@pytest.mark.skipif(not oracle_lib.have_ref(), reason="oracle/_ref not shipped")
@pytest.mark.parametrize("sid,algo,res,nit", [(1, 4, 128, 2), (3, 4, 128, 1), (0, 2, 96, 1), (2, 1, 96, 2), (1, 3, 96, 1)])
@pytest.mark.parametrize("strict", [False, True], ids=["wavefront", "strict"])
def test_hip_equals_unmodified_reference(sid, algo, res, nit, strict):
    """The GPU's tape replayed into the unmodified reference build."""
    mask = SCENE_CONFIGS[sid]
    sc = cornell_scene(sid, res, res)
    r = VertexCM(sc, algo, 0.003, 0.75, 1234, strict_order=strict)
    r.mMaxPathLength = 10
    lcs, ccs = [], []
    for it in range(nit):
        r.RunIteration(it)
        a, b = r.backend.rng_counts()
        lcs.append(a)
        ccs.append(b)
    fb, consumed, bad = oracle_lib.ref_run_tape(mask, res, res, algo, np.concatenate(lcs), np.concatenate(ccs), n_iter=nit)
    assert bad == 0, "reference consumed a different number of random floats than the GPU"
    _check_fb(r.framebuffer_sum(), fb, algo, strict)
    r.close()


def test_determinism_and_linearity():
    """vcm (splats included): two runs bit-identical; the framebuffer is a running sum."""
    sc = cornell_scene(1, 128, 128)
    a = VertexCM(sc, 4, 0.003, 0.75, 7)
    b = VertexCM(sc, 4, 0.003, 0.75, 7)
    a.mMaxPathLength = b.mMaxPathLength = 10
    a.RunIteration(0)
    b.RunIteration(0)
    f1 = a.framebuffer_sum()
    assert np.array_equal(f1, b.framebuffer_sum())
    a.RunIteration(1)
    f2 = a.framebuffer_sum()
    assert np.all(f2 >= f1) and f2.sum() > f1.sum()
    assert a.GetFramebuffer().mean() == pytest.approx(f2.mean() / 2, rel=1e-6)
    a.close()
    b.close()


def test_full_size_properties_2048():
    """BASELINE size (scene 1 vcm 2048^2): counters match the reference's
    measured workload within Monte-Carlo spread, image mean
    matches the known converged value, every random tape entry is in range."""
    sc = cornell_scene(1, 2048, 2048)
    r = VertexCM(sc, 4, 0.003, 0.75, 1234)
    r.mMaxPathLength = 10
    r.RunIteration(0)
    st = r.stats()
    assert abs(st["lightVertices"] - 8920850) < 0.01 * 8920850
    assert abs(st["mergeAccepted"] - 408257209) < 0.02 * 408257209
    assert abs(st["connections"] - 19617215) < 0.01 * 19617215
    fb = r.framebuffer_sum()
    assert np.isfinite(fb).all() and fb.min() >= 0
    assert np.allclose(fb.mean(axis=(0, 1)), [0.263, 0.266, 0.261], atol=0.01)
    lc, cc = r.backend.rng_counts()
    assert lc.min() >= 5 and lc.max() <= 5 + 9 * 4 and cc.min() >= 2 and cc.max() <= 2 + 10 * 7
    r.close()


FULL_FRAME = [(1, 4, 512, 0, True), (3, 4, 1024, 0, True), (1, 2, 2048, 0, False), (1, 4, 2048, 0, True), (1, 4, 2048, 25, False)]


@pytest.mark.parametrize("sid,algo,res,iteration,replay", FULL_FRAME,
                         ids=["C1-s1-vcm-512", "C2-s3-vcm-1024", "C3-s1-bpm-2048", "C4-s1-vcm-2048", "C4-s1-vcm-2048-iteration25"])
def test_baseline_configs_full_frame(sid, algo, res, iteration, replay):
    """Every GPU config of BASELINE.json at its FULL size, the WHOLE frame, bit for bit:
      * against the oracle (all host cores: light pass, grid build, every camera path) -- framebuffer, random-number
        tape of both passes and all workload counters;
      * `replay`: the GPU's tape replayed into the UNMODIFIED reference (oracle/_ref, serial like the reference:
        about 2 s at 512^2, a minute or so at 2048^2) -- framebuffer, and the reference must consume exactly the
        taped number of floats.
    Iteration 25 only sets the radius (vertexcm.hxx:295-296): more than 256 grid cells per axis at 2048^2, i.e. the
    coarsened query buckets."""
    import os
    sc = cornell_scene(sid, res, res)
    r = VertexCM(sc, algo, 0.003, 0.75, 1234)
    r.mMaxPathLength, r.mMinPathLength = 10, 0
    r.RunIteration(iteration)
    fb = r.framebuffer_sum()
    st = r.stats()
    lc, cc = r.backend.rng_counts()
    r.close()
    o = oracle_lib.Oracle(sc, algo, threads=os.cpu_count() or 1)
    o.run_iteration(iteration, 0, 10)
    ref = o.framebuffer()
    ost = o.stats()
    olc, occ = o.counts()
    del o
    assert np.array_equal(lc, olc) and np.array_equal(cc, occ), "random-number tape"
    for k in ("lightVertices", "lightRays", "cameraRays", "shadowRays", "mergeQueries", "mergeCandidates", "mergeAccepted",
              "connections", "lightSplats"):
        assert st[k] == ost[k], (k, st[k], ost[k])
    assert np.array_equal(fb.view(np.uint32), ref.view(np.uint32))
    assert fb.max() > 0
    if replay and oracle_lib.have_ref():
        rfb, consumed, bad = oracle_lib.ref_run_tape(SCENE_CONFIGS[sid], res, res, algo, lc, cc, first_iteration=iteration)
        assert bad == 0 and consumed == int(lc.sum(dtype=np.int64)) + int(cc.sum(dtype=np.int64))
        assert np.array_equal(fb.view(np.uint32), rfb.view(np.uint32))


# Box masks outside g_SceneConfigs (scene.hxx:112-126): vcm_scene_cornell accepts them, so they are rendered too
OTHER_MASKS = [(1 | 32, 4, 160, 2, 0, 10), (1 | 64 | 128, 4, 128, 1, 0, 10), (1 | 2 | 4 | 8 | 256 | 128, 4, 128, 2, 0, 10),
               (1 | 2 | 4 | 8 | 256 | 128, 2, 96, 1, 0, 10), (16 | 32 | 1 | 256, 3, 96, 1, 0, 10), (8 | 64, 5, 96, 1, 0, 10)]


@pytest.mark.parametrize("mask,algo,res,nit,mn,mx", OTHER_MASKS,
                         ids=["large-glass-sphere-diffuse-floor", "kDefault", "all-light-types-vcm", "all-light-types-bpm",
                              "both-large-spheres-bpt", "background-only-pt"])
def test_other_box_masks_equal_oracle_and_reference(mask, algo, res, nit, mn, mx):
    sc = cornell_scene(mask, res, res, is_mask=True)
    o = Oracle(sc, algo, threads=8)
    r = VertexCM(sc, algo, 0.003, 0.75, 1234)
    r.mMinPathLength, r.mMaxPathLength = mn, mx
    lcs, ccs = [], []
    for it in range(nit):
        o.run_iteration(it, mn, mx)
        r.RunIteration(it)
        lc, cc = r.backend.rng_counts()
        olc, occ = o.counts()
        assert np.array_equal(lc, olc) and np.array_equal(cc, occ)
        lcs.append(lc)
        ccs.append(cc)
        so, sg = o.stats(), r.stats()
        for k in ("lightVertices", "lightRays", "cameraRays", "shadowRays", "mergeQueries", "mergeCandidates",
                  "mergeAccepted", "connections", "lightSplats"):
            assert so[k] == sg[k], (k, so[k], sg[k])
    fb = r.framebuffer_sum()
    r.close()
    assert np.array_equal(fb.view(np.uint32), o.framebuffer().view(np.uint32))
    assert fb.max() > 0
    if oracle_lib.have_ref():
        rfb, consumed, bad = oracle_lib.ref_run_tape(mask, res, res, algo, np.concatenate(lcs), np.concatenate(ccs),
                                                     n_iter=nit, min_len=mn, max_len=mx)
        assert bad == 0
        assert np.array_equal(fb.view(np.uint32), rfb.view(np.uint32))


@pytest.mark.parametrize("mx,res", [(18, 96), (24, 128), (31, 64), (40, 48)])
def test_long_paths(mx, res):
    """maxPathLength 18..31 in wavefront mode: a camera vertex connects to up to maxPathLength - 2 light vertices, more
    than 15 per lane and more than one queue block per wave (wave_queue_alloc); 40 runs in the path (per-path masks are
    32 bits).  Glass sphere on a diffuse floor under the ceiling light: long specular chains."""
    sc = cornell_scene(1 | 32, res, res, is_mask=True)
    o = Oracle(sc, 4, threads=8)
    r = VertexCM(sc, 4, 0.003, 0.75, 1234)
    r.mMinPathLength, r.mMaxPathLength = 0, mx
    assert r.backend.L.vcm_is_wavefront(r.backend.ctx, mx) == (1 if mx <= 31 else 0)
    for it in range(2):
        o.run_iteration(it, 0, mx)
        r.RunIteration(it)
        so, sg = o.stats(), r.stats()
        for k in ("lightVertices", "connections", "shadowRays", "mergeAccepted"):
            assert so[k] == sg[k], (k, so[k], sg[k])
    lc, _ = r.backend.rng_counts()
    assert int(lc.max()) > 5 + 4 * 15, "no light path long enough to exercise the case"
    assert np.array_equal(r.framebuffer_sum().view(np.uint32), o.framebuffer().view(np.uint32))
    r.close()


# ---- the reference's two other renderers on the GPU (SURVEY section 8(f) "next" #2) ----------------------------------
SIMPLE_CASES = [(sid, algo, 96, 3, 0, 10) for sid in range(4) for algo in (5, 6)] + [
    (1, 5, 200, 2, 0, 10), (0, 5, 64, 1, 2, 5), (3, 5, 64, 1, 0, 1), (1, 5, 130, 1, 0, 3), (2, 5, 8, 1, 0, 10)]


@pytest.mark.parametrize("sid,algo,res,nit,mn,mx", SIMPLE_CASES)
def test_path_tracer_and_eye_light_equal_oracle_and_reference(sid, algo, res, nit, mn, mx):
    """PathTracer::RunIteration (pathtracer.hxx:45-215) and EyeLight::RunIteration (eyelight.hxx:46-77) through the
    same C-ABI (VCM_ALGO_PATH_TRACE / VCM_ALGO_EYE_LIGHT): tape, counters and framebuffer bit-exact against the
    oracle, and the GPU's tape replayed into the unmodified reference classes gives the same image."""
    sc = cornell_scene(sid, res, res)
    o = Oracle(sc, algo, threads=8)
    r = VertexCM(sc, algo, 0.003, 0.75, 1234)
    r.mMinPathLength, r.mMaxPathLength = mn, mx
    lcs, ccs = [], []
    for it in range(nit):
        o.run_iteration(it, mn, mx)
        r.RunIteration(it)
        lc, cc = r.backend.rng_counts()
        olc, occ = o.counts()
        assert lc.max() == 0 and np.array_equal(cc, occ), "camera tape"
        lcs.append(lc)
        ccs.append(cc)
        so, sg = o.stats(), r.stats()
        for k in ("lightVertices", "cameraRays", "shadowRays", "mergeQueries", "connections", "lightSplats"):
            assert so[k] == sg[k], (k, so[k], sg[k])
    fb = r.framebuffer_sum()
    assert np.array_equal(fb.view(np.uint32), o.framebuffer().view(np.uint32))
    assert fb.max() > 0 or mx == 1   # maxPathLength 1 sees only directly visible emitters: none in scene 3
    if oracle_lib.have_ref():
        ref, consumed, bad = oracle_lib.ref_run_tape(SCENE_CONFIGS[sid], res, res, algo, np.concatenate(lcs),
                                                     np.concatenate(ccs), n_iter=nit, min_len=mn, max_len=mx)
        assert bad == 0 and consumed == int(np.concatenate(ccs).sum())
        assert np.array_equal(fb.view(np.uint32), ref.view(np.uint32))
    r.close()


def test_path_tracer_converges_to_the_vcm_image():
    """Independent estimators of the same image (the cross-check SURVEY 8(f) asks the GPU path tracer for)."""
    sc = cornell_scene(1, 128, 128)
    imgs = {}
    for algo, n in ((5, 64), (4, 24)):
        r = VertexCM(sc, algo, 0.003, 0.75, 1234)
        r.mMaxPathLength = 10
        for it in range(n):
            r.RunIteration(it)
        imgs[algo] = r.GetFramebuffer()
        r.close()
    a, b = imgs[5].mean(axis=(0, 1)), imgs[4].mean(axis=(0, 1))
    assert np.all(np.abs(a - b) < 0.03 * b), (a, b)


@pytest.mark.parametrize("sid,algo,res", [(1, 4, 128), (3, 2, 96), (0, 4, 64)])
def test_late_iterations_equal_oracle(sid, algo, res):
    """aIteration only drives the radius (vertexcm.hxx:295-296): late iterations have small radii, i.e. more grid cells
    per axis than the query-sort bucket table has entries per axis -- the coarsened-bucket path of K4a -- and, at
    iteration 300, a radius near the 1e-7 clamp of :298 is still far; the renderer-local counter wraps the 64-entry
    statistics ring."""
    sc = cornell_scene(sid, res, res)
    o = Oracle(sc, algo, threads=8)
    r = VertexCM(sc, algo, 0.003, 0.75, 1234)
    r.mMaxPathLength = 10
    for it in (9, 40, 300, 5000):
        o.run_iteration(it, 0, 10)
        r.RunIteration(it)
        so, sg = o.stats(), r.stats()
        for k in ("lightVertices", "mergeQueries", "mergeCandidates", "mergeAccepted", "connections"):
            assert so[k] == sg[k], (it, k, so[k], sg[k])
        assert abs(sg["radius"] - so["radius"]) == 0
    assert np.array_equal(r.framebuffer_sum().view(np.uint32), o.framebuffer().view(np.uint32))
    r.close()


def test_statistics_ring_keeps_the_last_iterations():
    sc = cornell_scene(1, 64, 64)
    r = VertexCM(sc, 4, 0.003, 0.75, 1234)
    r.mMaxPathLength = 10
    per_iter = []
    for it in range(70):            # more than the 64 slots of the ring
        r.RunIteration(it)
        if it >= 60:
            per_iter.append(r.stats())
    hist = [r.backend.stats_at(ago) for ago in range(10)]   # read after the batch: ago 0 = iteration 69
    for ago, h in enumerate(hist):
        ref = per_iter[9 - ago]
        for k in ("lightVertices", "mergeCandidates", "mergeAccepted", "connections", "radius"):
            assert h[k] == ref[k], (ago, k)
        assert h["msTotal"] > 0 and h["msMergeKernel"] > 0
    with pytest.raises(RuntimeError):
        r.backend.stats_at(64)
    r.close()


def test_soak_many_iterations_stay_exact():
    """40 consecutive iterations (side-stream grid build, statistics ring, shrinking radius, two renderers
    interleaved on the shared arena): the accumulated framebuffer must still equal the oracle's bit for bit."""
    import os
    sc = cornell_scene(1, 320, 240)
    o = Oracle(sc, 4, threads=os.cpu_count() or 1)
    r = VertexCM(sc, 4, 0.003, 0.75, 1234)
    other = VertexCM(cornell_scene(3, 200, 200), 2, 0.003, 0.75, 99)   # a second renderer borrowing the same arena
    r.mMaxPathLength = other.mMaxPathLength = 10
    for it in range(40):
        o.run_iteration(it, 0, 10)
        r.RunIteration(it)
        if it % 3 == 0:
            other.RunIteration(it)
    assert np.array_equal(r.framebuffer_sum().view(np.uint32), o.framebuffer().view(np.uint32))
    so, sg = o.stats(), r.stats()
    for k in ("lightVertices", "mergeCandidates", "mergeAccepted", "connections", "lightSplats"):
        assert so[k] == sg[k], k
    o2 = Oracle(cornell_scene(3, 200, 200), 2, seed=99, threads=os.cpu_count() or 1)
    for it in range(0, 40, 3):
        o2.run_iteration(it, 0, 10)
    assert np.array_equal(other.framebuffer_sum().view(np.uint32), o2.framebuffer().view(np.uint32))
    r.close()
    other.close()


def test_single_rank_local_bbox_before_build_grid():
    """vcm_local_light_bbox on a single-rank context overwrites the key words K1 left in the grid header; the grid
    build that follows must take the finalised box, not finalise the words a second time."""
    sc = cornell_scene(1, 96, 96)
    o = Oracle(sc, 4, threads=8)
    r = VertexCM(sc, 4, 0.003, 0.75, 1234)
    b = r.backend
    for it in range(2):
        o.run_iteration(it, 0, 10)
        b.begin(it, 0, 10)
        b.trace_light()
        mn, mx, n = b.local_bbox()
        b.build_grid()
        b.trace_camera()
        b.merge()
        b.end()
        ce, idx, bbox = o.grid()
        cs, sidx, gb = b.grid()
        assert n == o.stats()["lightVertices"]
        assert np.array_equal(np.array(mn + mx, np.float32), bbox)
        assert np.array_equal(gb, bbox)
        assert np.array_equal(cs[1:], ce) and np.array_equal(sidx, idx)
    assert np.array_equal(b.framebuffer_sum(), o.framebuffer())
    r.close()


def test_misordered_end_iteration_is_refused_and_the_iteration_stays_open():
    """vcm_end_iteration before vcm_merge is a host error, not a device failure: it returns -1 and leaves the iteration as
    it is -- the light splats already applied belong to an iteration that can still be finished and counted (ADVICE r3:
    tearing it down left them in the framebuffer with the iteration uncounted)."""
    L = load_library()
    sc = cornell_scene(1, 32, 32)
    r = VertexCM(sc, 4, 0.003, 0.75, 1234)
    b = r.backend
    b.begin(0, 0, 10)
    b.trace_light()
    assert L.vcm_end_iteration(b.ctx) != 0          # vcm_merge has not run
    assert b"vcm_merge" in L.vcm_last_error()
    assert L.vcm_iterations(b.ctx) == 0
    b.build_grid()                                   # ... and the host carries on
    b.trace_camera()
    b.merge()
    b.end()
    assert L.vcm_iterations(b.ctx) == 1
    o = Oracle(sc, 4, threads=4)
    o.run_iteration(0, 0, 10)
    assert np.array_equal(b.framebuffer_sum(), o.framebuffer())
    r.close()


@pytest.mark.parametrize("algo", [0, 4])
def test_framebuffer_access_inside_an_iteration_waits_for_the_light_splats(algo):
    """The light splats (K1c / K1d) run on a stream of their own from vcm_trace_light on; every accessor of the framebuffer
    joins that stream first.  Light tracing: what is read right after vcm_trace_light is the whole image of
    the iteration; a clear at that point removes all of it."""
    sc = cornell_scene(1, 96, 96)
    o = Oracle(sc, algo, threads=4)
    o.run_iteration(0, 0, 10)
    lt = Oracle(sc, 0, threads=4)
    lt.run_iteration(0, 0, 10)
    for clear in (False, True):
        r = VertexCM(sc, algo, 0.003, 0.75, 1234)
        b = r.backend
        b.begin(0, 0, 10)
        b.trace_light()
        mid = b.framebuffer_sum()                    # joins the splat stream
        if algo == 0:
            assert np.array_equal(mid, o.framebuffer())
        else:   # VCM: the MIS-weighted light image so far, complete or not started -- never torn
            assert np.array_equal(mid, np.zeros_like(mid)) or mid.sum() > 0
        if clear:
            b.clear_framebuffer()
        b.build_grid()
        b.trace_camera()
        b.merge()
        b.end()
        fin = b.framebuffer_sum()
        if algo == 0:
            assert np.array_equal(fin, np.zeros_like(fin) if clear else o.framebuffer())
        elif not clear:
            assert np.array_equal(fin, o.framebuffer())
        r.close()
