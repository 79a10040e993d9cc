"""Generates the committed golden fixtures from the UNMODIFIED reference
(/root/reference, through oracle/_ref built by oracle/Makefile).  Run in the
build container only (the GPU box has no /root/reference):

    python tests/golden/gen_golden.py

Fixtures
  scene_<id>_<res>.bin      vcm_scene_desc flattened from the reference's own
                            Scene (LoadCornellBox + BuildSceneSphere + Camera)
  fb_goldens.npz            raw fp32 framebuffer sums of the reference's
                            VertexCM::RunIteration for every scene x algorithm
                            (32x32, 2 iterations, default path lengths) plus a
                            few path-length windows, rendered with the
                            tape-replay Rng; the tape (per-path float counts)
                            is stored next to each image.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle_lib import Oracle, ref_scene, ref_run_tape  # noqa: E402
from smallvcm_amd._abi import SCENE_CONFIGS  # noqa: E402

SCENE_RES = [16, 32, 64, 512, 1024, 2048]
FB_CASES = [(sid, algo, 32, 2, 0, 10) for sid in range(4) for algo in range(5)] + [
    (1, 4, 48, 1, 2, 6), (3, 4, 48, 1, 0, 3), (0, 2, 48, 1, 4, 10), (2, 3, 24, 3, 0, 10), (1, 4, 16, 1, 0, 1)]


def main():
    for sid in range(4):
        for res in SCENE_RES:
            d = ref_scene(SCENE_CONFIGS[sid], res, res)
            with open(os.path.join(HERE, "scene_%d_%d.bin" % (sid, res)), "wb") as f:
                f.write(d.tobytes())
    out = {}
    for (sid, algo, res, nit, mn, mx) in FB_CASES:
        mask = SCENE_CONFIGS[sid]
        sc = ref_scene(mask, res, res)
        o = Oracle(sc, algo)
        lcs, ccs = [], []
        for it in range(nit):
            o.run_iteration(it, mn, mx)
            a, b = o.counts()
            lcs.append(a)
            ccs.append(b)
        lc, cc = np.concatenate(lcs), np.concatenate(ccs)
        fb, consumed, bad = ref_run_tape(mask, res, res, algo, lc, cc, n_iter=nit, min_len=mn, max_len=mx)
        assert bad == 0, "reference desynchronised from the tape"
        assert np.array_equal(fb, o.framebuffer()), "oracle != reference; fix the oracle first"
        key = "s%d_a%d_r%d_i%d_l%d_%d" % (sid, algo, res, nit, mn, mx)
        out[key + "_fb"] = fb
        out[key + "_lc"] = lc
        out[key + "_cc"] = cc
    np.savez_compressed(os.path.join(HERE, "fb_goldens.npz"), **out)
    print("wrote", len(FB_CASES), "framebuffer goldens and", 4 * len(SCENE_RES), "scene descs")


if __name__ == "__main__":
    main()
