#!/usr/bin/env python
"""TEST INFRASTRUCTURE.  north_star: "per-pixel RMSE < 1e-4 vs reference at fixed seed".  What is left between the
product's definition of sinf / cosf / powf (smallvcm_amd/csrc/detmath.h = oracle/detmath_ref.h) and the reference AS
SHIPPED, which calls the image's libm (src/utils.hxx:85-117, :173-199, src/bsdf.hxx:290-318, :414-446)?

Since round 4 detmath RESTATES that libm (glibc 2.35, FMA variants; oracle/libm_check.c pins it over all 2^32
arguments) with one deviation: an integer exponent -- the Phong lobe's x^90 -- is the correctly rounded power, which
glibc's own powf misses by one ulp for 0.17 % of the arguments.  Two measurements per BASELINE configuration, one
iteration (iteration 0, seed 1234, maxPathLength 10), same random numbers:

    replay   the UNMODIFIED reference with ITS OWN libm (oracle/_ref/libsmallvcm_ref_tape_libm.so: nothing interposed)
             replays the oracle's tape, serially, exactly as src/vertexcm.hxx:284-548 runs: desynchronisation flag,
             floats consumed, per-pixel RMSE / max |delta| / pixels that differ at all;
    oracle   the oracle built over the host's libm calls (liboracle_glibc.so) and over round 1's "evaluate in binary64,
             round once" (liboracle_cr.so) against the checker (liboracle.so): paths whose random-float count differs,
             RMSE.

    make -C oracle glibc && python oracle/libm_tolerance.py [--full]   -> profiles/archive/r06_libm_tolerance.json
(--full adds the two 2048^2 configurations C3 and C4: the serial reference takes a few minutes for each.)
Rounds 2-3 (binary32 polynomials, <= 1.9 ulp) measured RMSE 2.2e-4 at C1 and 7.9e-5 at C4 with this tool
(profiles/archive/r05_libm_tolerance.json)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib  # noqa: E402
from smallvcm_amd._abi import SCENE_CONFIGS  # noqa: E402

VARIANTS = {"det": "liboracle.so", "glibc": "liboracle_glibc.so", "cr": "liboracle_cr.so"}


def render(lib, sc, algo, iteration, threads):
    o = oracle_lib.Oracle(sc, algo, threads=threads, lib=lib)
    o.run_iteration(iteration, 0, 10)
    lc, cc = o.counts()
    return o.framebuffer().astype(np.float64), lc, cc


def image_delta(fa, fb):
    d = fa - fb
    pix = np.abs(d).max(axis=2)
    mean = float(fa.mean())
    return {"rmse": float(np.sqrt((d ** 2).mean())), "max_abs": float(np.abs(d).max()),
            "differing_pixel_fraction": float((pix > 0).mean()), "image_mean": mean,
            "rmse_over_mean": float(np.sqrt((d ** 2).mean()) / mean), "mean_shift_over_mean": float(abs(d.mean()) / mean)}


def run(name, scene, algo, res, threads):
    mask = SCENE_CONFIGS[scene]
    sc = oracle_lib.ref_scene(mask, res, res)
    libs = {k: oracle_lib.load_oracle(os.path.join(ROOT, "oracle", v)) for k, v in VARIANTS.items()}
    r = {k: render(L, sc, algo, 0, threads) for k, L in libs.items()}
    fb, lc, cc = r["det"]
    out = {"config": "%s: scene %d algorithm %d %dx%d iteration 0 seed 1234 maxPathLength 10" % (name, scene, algo, res, res),
           "paths": int(lc.size + cc.size), "random_floats": int(lc.sum()) + int(cc.sum())}
    t = time.time()
    ref, consumed, bad = oracle_lib.ref_run_tape(mask, res, res, algo, lc, cc, own_libm=True)
    out["replay_unmodified_reference_own_libm"] = dict(image_delta(fb, ref.astype(np.float64)), desynchronised=int(bad),
                                                       floats_consumed=int(consumed), seconds=round(time.time() - t, 1))
    for k in ("glibc", "cr"):
        fk, lk, ck = r[k]
        out["oracle_det_vs_" + k] = dict(image_delta(fb, fk), paths_with_different_float_count=int((lk != lc).sum() + (ck != cc).sum()))
    return out


def main():
    threads = oracle_lib.oracle().oracle_max_threads()
    out = {"note": "see oracle/libm_tolerance.py", "host_threads": threads, "glibc": os.confstr("CS_GNU_LIBC_VERSION"), "configs": []}
    cfgs = [("C1", 1, 4, 512), ("C2", 3, 4, 1024)]
    if "--full" in sys.argv:
        cfgs += [("C3", 1, 2, 2048), ("C4", 1, 4, 2048)]
    for c in cfgs:
        r = run(*c, threads=threads)
        out["configs"].append(r)
        p = r["replay_unmodified_reference_own_libm"]
        print("%s replay of the unmodified reference, its own libm: desync %d, floats %d / %d, RMSE %.3e (%.2e of the mean), max |d| %.3e, "
              "pixels that differ %.2e  [%.0f s]" % (c[0], p["desynchronised"], p["floats_consumed"], r["random_floats"], p["rmse"],
                                                     p["rmse_over_mean"], p["max_abs"], p["differing_pixel_fraction"], p["seconds"]), flush=True)
        for k in ("glibc", "cr"):
            q = r["oracle_det_vs_" + k]
            print("%s oracle det vs %-5s: paths with another float count %d / %d, RMSE %.3e, max |d| %.3e, pixels that differ %.2e"
                  % (c[0], k, q["paths_with_different_float_count"], r["paths"], q["rmse"], q["max_abs"], q["differing_pixel_fraction"]), flush=True)
        json.dump(out, open(os.path.join(ROOT, "profiles", "r06_libm_tolerance.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
