"""The order-relaxed mode (vcm_set_relaxed_order, round 5): every addend of the default mode, added to its pixel with an fp32
atomic instead of in the order of the reference's serial loops (src/vertexcm.hxx:321-396, :415-545, src/framebuffer.hxx:43-57).
Same work (all counters equal), same addends -- the frame may differ from the bit-exact default (= the reference's, see
test_gpu_parity.py) by the rounding of a different summation order only.  north_star's bar is RMSE < 1e-4; asserted here: 1e-6
of the frame's mean, and no pixel further than a few ulps of its own magnitude.  GPU only."""
import numpy as np
import pytest

from smallvcm_amd.renderer import VertexCM, cornell_scene

pytestmark = pytest.mark.gpu

ALGO = {"lt": 0, "ppm": 1, "bpm": 2, "bpt": 3, "vcm": 4}


def _render(scene, algo, res, iters, relaxed, max_len=10, min_len=0):
    r = VertexCM(cornell_scene(scene, res, res), ALGO[algo], 0.003, 0.75, 1234)
    r.mMaxPathLength, r.mMinPathLength = max_len, min_len
    if relaxed:
        r.backend.set_relaxed_order(True)
    st = []
    for it in range(iters):
        r.RunIteration(it)
        st.append(r.stats())
    fb = r.framebuffer_sum().astype(np.float64)
    r.close()
    return fb, st


@pytest.mark.parametrize("scene,algo,res,iters", [(1, "vcm", 512, 2), (3, "vcm", 256, 2), (1, "bpm", 256, 2), (1, "bpt", 256, 2),
                                                   (1, "lt", 256, 2), (2, "ppm", 192, 2), (0, "vcm", 192, 3)])
def test_relaxed_order_differs_from_the_default_by_rounding_only(scene, algo, res, iters):
    ref, st0 = _render(scene, algo, res, iters, False)
    fb, st1 = _render(scene, algo, res, iters, True)
    for a, b in zip(st0, st1):   # the same paths, vertices, rays, candidates, accepted photons, connections, splats
        for k in ("lightVertices", "lightRays", "cameraRays", "shadowRays", "mergeQueries", "mergeCandidates", "mergeAccepted",
                  "connections", "lightSplats"):
            assert a[k] == b[k], k
    rmse = float(np.sqrt(np.mean((fb - ref) ** 2)))
    assert rmse < 1e-6 * max(ref.mean(), 1e-3), (rmse, ref.mean())
    # per pixel: a sum of n fp32 addends in another order moves by a few ulps of the largest partial sum
    tol = 64 * np.finfo(np.float32).eps * np.maximum(np.abs(ref), ref.mean()) + 1e-12
    assert np.all(np.abs(fb - ref) <= tol), float(np.max(np.abs(fb - ref) / tol))


def test_relaxed_order_with_a_path_length_window_and_the_wrong_renderers():
    ref, _ = _render(1, "vcm", 192, 2, False, max_len=6, min_len=3)
    fb, _ = _render(1, "vcm", 192, 2, True, max_len=6, min_len=3)
    assert float(np.sqrt(np.mean((fb - ref) ** 2))) < 1e-6 * ref.mean()
    r = VertexCM(cornell_scene(1, 64, 64), 5, 0.003, 0.75, 1234)   # PathTracer: one addition per path, nothing to relax
    with pytest.raises(RuntimeError):
        r.backend.set_relaxed_order(True)
    r.close()
    r = VertexCM(cornell_scene(1, 64, 64), 4, 0.003, 0.75, 1234, strict_order=True)
    with pytest.raises(RuntimeError):
        r.backend.set_relaxed_order(True)
    r.close()
