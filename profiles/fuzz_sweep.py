#!/usr/bin/env python
"""Random scenes beyond the seeds tests/test_fuzz_scenes.py holds: device vs oracle (tape, counters, framebuffer, bit for
bit) for as many seeds as fit in --seconds.  Also times the framebuffer read-out a host would pay if it fetched the image
after every iteration (the PCIe-inclusive rate DESIGN.md quotes).   python profiles/fuzz_sweep.py --first 24 --seconds 240"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_lib import Oracle  # noqa: E402
from mesh_scenes import random_scene  # noqa: E402
from smallvcm_amd.renderer import VertexCM  # noqa: E402

COUNTERS = ("lightVertices", "lightRays", "cameraRays", "shadowRays", "mergeQueries", "mergeCandidates", "mergeAccepted",
            "connections", "lightSplats")


def one(seed, res, algo):
    sc = random_scene(seed, res, res)
    o = Oracle(sc, algo, threads=os.cpu_count() or 1)
    r = VertexCM(sc, algo, 0.003, 0.75, 1234)
    r.mMaxPathLength = 10
    ok = True
    for it in range(2):
        o.run_iteration(it, 0, 10)
        r.RunIteration(it)
        lc, cc = r.backend.rng_counts()
        olc, occ = o.counts()
        ok &= bool(np.array_equal(lc, olc) and np.array_equal(cc, occ))
        so, sg = o.stats(), r.stats()
        ok &= all(so[k] == sg[k] for k in COUNTERS)
    fb = r.framebuffer_sum()
    r.close()
    ok &= bool(np.array_equal(fb.view(np.uint32), o.framebuffer().view(np.uint32)))
    return ok, sc.nPrims, so["mergeCandidates"], so["connections"]


def readout():
    import ctypes as C
    from smallvcm_amd.renderer import cornell_scene
    sc = cornell_scene(1, 2048, 2048)
    out = np.zeros((2048, 2048, 3), np.float32)   # the host's image, allocated (and its pages touched) once
    fp = out.ctypes.data_as(C.POINTER(C.c_float))
    r = VertexCM(sc, 4, 0.003, 0.75, 1234)
    r.mMaxPathLength = 10
    for it in range(5):
        r.RunIteration(it)
    r.framebuffer_sum()
    t = time.time()
    for it in range(5, 25):
        r.RunIteration(it)
    r.backend.synchronize()
    a = (time.time() - t) / 20
    t = time.time()
    for it in range(25, 45):
        r.RunIteration(it)
        assert r.backend.L.vcm_read_framebuffer(r.backend.ctx, fp) == 0
    b = (time.time() - t) / 20
    r.close()
    paths = 2 * 2048 * 2048
    print("C4 through the Python mirror: %.2f ms per iteration (%.0f Mpaths/s); with the 50 MB framebuffer read to the host after EVERY "
          "iteration %.2f ms (%.0f Mpaths/s)" % (a * 1e3, paths / a / 1e6, b * 1e3, paths / b / 1e6), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--first", type=int, default=24)
    ap.add_argument("--seconds", type=float, default=240)
    ap.add_argument("--readout", action="store_true")
    a = ap.parse_args()
    if a.readout:
        readout()
    t0, seed, bad, n = time.time(), a.first, [], 0
    while time.time() - t0 < a.seconds:
        res = (64, 96, 128, 200)[seed % 4]
        for algo in (4, (2, 3, 5, 1, 0, 6)[seed % 6]):
            ok, prims, cand, conn = one(seed, res, algo)
            n += 1
            if not ok:
                bad.append((seed, res, algo))
                print("MISMATCH seed %d res %d algorithm %d (%d primitives)" % (seed, res, algo, prims), flush=True)
        seed += 1
    print("random scenes: seeds %d..%d, %d renders of two iterations each, %d mismatches %s  [%.0f s]"
          % (a.first, seed - 1, n, len(bad), bad, time.time() - t0), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
