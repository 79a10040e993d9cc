#!/usr/bin/env python
"""Where the wave time of the big kernels goes (measurement build only).

    make -C smallvcm_amd/csrc variant NAME=rc EXTRA=-DVCM_REGION_CLOCK
    SMALLVCM_AMD_LIB=smallvcm_amd/csrc/libsmallvcm_amd_rc.so python profiles/tools/region_clock.py [scene] [res] [algo]

The variant brackets regions of light_path_step / camera_path_step / the task evaluations / merge_query_walk with
s_memtime (vcm_core.h RC_MARK); a wave charges the shader-clock cycles since its previous mark to the region.  The
figure is wave RESIDENCE time (issue + waiting + the other waves of the SIMD), summed over the waves: shares inside
one kernel say where its time goes; the absolute values of different kernels compare only through their occupancy."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from smallvcm_amd import renderer as R  # noqa: E402

NAMES = {
    19: ("K1", "refill + light_path_begin"), 0: ("K1", "scene_intersect"), 1: ("K1", "bsdf_setup + MIS"),
    2: ("K1", "store the vertex"), 3: ("K1", "sample_scattering"), 21: ("K1", "end of step (path end stores)"),
    20: ("K3", "refill + camera_path_begin"), 4: ("K3", "scene_intersect"), 5: ("K3", "bsdf_setup + MIS + emission"),
    6: ("K3", "append vertex + DI / VC tasks"), 7: ("K3", "sample_scattering"), 22: ("K3", "end of step"),
    17: ("K1c", "connect_to_camera: evaluation"), 18: ("K1c", "connect_to_camera: scene_occluded"),
    9: ("K3b", "direct_illumination: evaluation"), 10: ("K3b", "direct_illumination: scene_occluded"),
    12: ("K3c", "connect_vertices: evaluation"), 13: ("K3c", "connect_vertices: scene_occluded"),
    23: ("K1 sample_scattering", "rng_peek (Philox)"), 24: ("K1 sample_scattering", "bsdf_sample"), 25: ("K1 sample_scattering", "bsdf_pdf (reverse)"),
    26: ("K1 sample_scattering", "Russian roulette + MIS update"),
    27: ("K3 sample_scattering", "rng_peek (Philox)"), 28: ("K3 sample_scattering", "bsdf_sample"), 29: ("K3 sample_scattering", "bsdf_pdf (reverse)"),
    30: ("K3 sample_scattering", "Russian roulette + MIS update"),
    32: ("K1+K3 bsdf_sample", "event pick, material, sincos"), 33: ("K1+K3 bsdf_sample", "diffuse branch"), 34: ("K1+K3 bsdf_sample", "Phong branch"),
    35: ("K1+K3 bsdf_sample", "mirror branch"), 36: ("K1+K3 bsdf_sample", "refraction branch"),
    38: ("K1+K3 scene_intersect", "certified filter over the list"), 39: ("K1+K3 scene_intersect", "the winner's exact arithmetic (triangle)"),
    41: ("K1+K3 scene_intersect", "the winner's exact arithmetic (sphere, binary64)"), 40: ("K1+K3 scene_intersect", "exact list walk (a lane was uncertain)"),
    14: ("K4", "cells + set-up"), 15: ("K4", "scan"), 16: ("K4", "drain (RangeQuery::Process)"),
}


def main():
    scene_id = sys.argv[1] if len(sys.argv) > 1 else "1"   # a built-in configuration, or the path of a .vcmscene file
    res = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    algo = {"vcm": R.VertexCM.kVcm, "bpm": R.VertexCM.kBpm, "bpt": R.VertexCM.kBpt}[sys.argv[3] if len(sys.argv) > 3 else "vcm"]
    L = R.load_library()
    L.region_clock_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    if scene_id.isdigit():
        sc = R.cornell_scene(int(scene_id), res, res)
    else:
        from smallvcm_amd.scene_file import load_scene
        sc = load_scene(scene_id, res, res)
    r = R.VertexCM(sc, algo, 0.003, 0.75, 1234)
    r.mMaxPathLength = 10
    for it in range(5):
        r.RunIteration(it)
    r.backend.synchronize()
    IDS = 64
    buf = (C.c_ulonglong * (3 * IDS))()
    assert L.region_clock_read(buf, 1) == 0
    n = 10
    for it in range(5, 5 + n):
        r.RunIteration(it)
    r.backend.synchronize()
    assert L.region_clock_read(buf, 0) == 0
    per = {}
    for rid, (k, name) in NAMES.items():
        per.setdefault(k, []).append((rid, name, buf[rid] / n, buf[IDS + rid] / n, buf[2 * IDS + rid] / n))
    print("scene %s %dx%d, mean of %d iterations; cycles are wave-residence shader clocks summed over waves" % (scene_id, res, res, n))
    for k, rows in per.items():
        tot = sum(x[2] for x in rows) or 1.0
        print("%s: %.1f M wave-cycles per iteration" % (k, tot / 1e6))
        lanes = sum(x[4] for x in rows)
        print("   (time-weighted lanes at the marks: %.2f of 64)" % (lanes / tot / 64.0))
        for rid, name, cyc, marks, lc in rows:
            print("   %-44s %6.1f %%  %9.0f marks  %8.0f cycles per mark  lanes %.2f" % (name, 100.0 * cyc / tot, marks, cyc / marks if marks else 0.0, lc / cyc / 64.0 if cyc else 0.0))
    r.close()


if __name__ == "__main__":
    main()
