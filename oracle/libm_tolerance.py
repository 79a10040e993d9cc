#!/usr/bin/env python
"""TEST INFRASTRUCTURE.  What does the deterministic sin / cos / pow (smallvcm_amd/csrc/detmath.h = oracle/detmath_ref.h)
cost against the arithmetic the reference itself is built with?

Every bit-exact claim of this repository is "GPU == reference LINKED AGAINST detmath" (oracle/ref_driver.cpp interposes
sinf / cosf / powf).  The reference as shipped calls the host's libm (src/utils.hxx:85-117, :173-199,
src/bsdf.hxx:290-318, :414-446).  Tape replay cannot compare the two -- one flipped decision desynchronises the serial
tape -- but the oracle can: its random numbers are keyed per path, so the SAME restatement built over three definitions
of the three functions renders the same paths, and whatever differs is the definitions' doing:

    det    liboracle.so        the specification (round 2: binary32, <= 1.6 / 1.9 ulp)
    glibc  liboracle_glibc.so  the host's sinf / cosf / powf  = the reference's own arithmetic
    cr     liboracle_cr.so     round 1's definition: evaluated in binary64, rounded once

Per configuration and pair: paths whose control flow differs (number of random floats consumed: a different
Russian-roulette / termination decision), pixels that differ at all, per-pixel RMSE and max |delta| of one iteration's
framebuffer, and the same relative to the image mean.

    make -C oracle glibc && python oracle/libm_tolerance.py [--full]   -> profiles/r05_libm_tolerance.json
(--full adds C4, scene 1 VCM 2048^2: minutes on 8 cores.)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib  # noqa: E402
from smallvcm_amd.renderer import cornell_scene  # noqa: E402

VARIANTS = {"det": "liboracle.so", "glibc": "liboracle_glibc.so", "cr": "liboracle_cr.so"}


def render(lib, scene, algo, res, iteration, threads):
    o = oracle_lib.Oracle(cornell_scene(scene, res, res), algo, threads=threads, lib=lib)
    o.run_iteration(iteration, 0, 10)
    lc, cc = o.counts()
    return o.framebuffer().astype(np.float64), lc.astype(np.int32), cc.astype(np.int32), o.stats()


def compare(a, b):
    fa, la, ca, sa = a
    fb, lb, cb, sb = b
    d = fa - fb
    n_paths = la.size + ca.size
    flipped = int((la != lb).sum() + (ca != cb).sum())
    pix = np.abs(d).max(axis=2)
    mean = float(fa.mean())
    return {"paths": int(n_paths), "paths_with_different_float_count": flipped, "flipped_path_fraction": flipped / n_paths,
            "pixels": int(pix.size), "pixels_that_differ": int((pix > 0).sum()), "differing_pixel_fraction": float((pix > 0).mean()),
            "rmse": float(np.sqrt((d ** 2).mean())), "max_abs": float(np.abs(d).max()), "image_mean": mean,
            "rmse_over_mean": float(np.sqrt((d ** 2).mean()) / mean), "mean_shift_over_mean": float(abs(d.mean()) / mean),
            "light_vertices": [int(sa["lightVertices"]), int(sb["lightVertices"])],
            "merges_accepted": [int(sa["mergeAccepted"]), int(sb["mergeAccepted"])]}


def run(name, scene, algo, res, iteration, threads):
    libs = {k: oracle_lib.load_oracle(os.path.join(ROOT, "oracle", v)) for k, v in VARIANTS.items()}
    r = {k: render(L, scene, algo, res, iteration, threads) for k, L in libs.items()}
    return {"config": "%s: scene %d algorithm %d %dx%d iteration %d seed 1234 maxPathLength 10" % (name, scene, algo, res, res, iteration),
            "det_vs_glibc": compare(r["det"], r["glibc"]), "cr_vs_glibc": compare(r["cr"], r["glibc"]),
            "det_vs_cr": compare(r["det"], r["cr"])}


def main():
    threads = oracle_lib.oracle().oracle_max_threads()
    out = {"note": __doc__.split("\n\n")[2], "host_threads": threads, "configs": []}
    cfgs = [("C1", 1, 4, 512, 0)]
    if "--full" in sys.argv:
        cfgs.append(("C4", 1, 4, 2048, 0))
    for c in cfgs:
        r = run(*c, threads=threads)
        out["configs"].append(r)
        for pair in ("det_vs_glibc", "cr_vs_glibc", "det_vs_cr"):
            p = r[pair]
            print("%s %-12s flipped paths %d / %d = %.2e   pixels that differ %.2e   RMSE %.3e (%.2e of the mean)   max |d| %.3e"
                  % (c[0], pair, p["paths_with_different_float_count"], p["paths"], p["flipped_path_fraction"],
                     p["differing_pixel_fraction"], p["rmse"], p["rmse_over_mean"], p["max_abs"]), flush=True)
    json.dump(out, open(os.path.join(ROOT, "profiles", "r05_libm_tolerance.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
