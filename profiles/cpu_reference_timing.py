"""One-off measurement on the GPU box's host cores (run through gpurun from the repo root):
the UNMODIFIED reference (oracle/_ref/libsmallvcm_ref_stock.so: src/*.hxx compiled as they lie, mt19937 + glibc,
render()'s iteration-parallel OpenMP loop) next to the oracle port (path-parallel) that bench.py's cpu_baseline
leg times.  Writes one JSON object; the numbers are quoted in DESIGN.md section 5."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib  # noqa: E402
from smallvcm_amd._abi import SCENE_CONFIGS  # noqa: E402
from smallvcm_amd.renderer import cornell_scene  # noqa: E402

cores = os.cpu_count()
out = {"host_cores": cores, "meminfo": open("/proc/meminfo").read().split("\n")[0]}
mask = SCENE_CONFIGS[1]


def stock(res, threads, iters):
    _, wall = oracle_lib.ref_render_stock(mask, res, res, 4, iterations=iters, threads=threads)
    return {"res": res, "threads": threads, "iterations": iters, "wall_s": round(wall, 3),
            "Mpaths_s": round(2.0 * res * res * iters / wall / 1e6, 4)}


def port(res, row_stride):
    sc = cornell_scene(1, res, res)
    o = oracle_lib.Oracle(sc, 4, threads=cores)
    t0 = time.perf_counter()
    o.begin(0, 0, 10); o.trace_light(); t1 = time.perf_counter()
    o.build_grid(); t2 = time.perf_counter()
    o.trace_camera(row_stride=row_stride); t3 = time.perf_counter()
    o.end()
    rows = len(range(0, res, row_stride))
    t_full = (t2 - t0) + (t3 - t2) * res / rows
    return {"res": res, "threads": cores, "row_stride": row_stride, "light_s": round(t1 - t0, 3), "grid_s": round(t2 - t1, 3),
            "camera_sample_s": round(t3 - t2, 3), "Mpaths_s": round(2.0 * res * res / t_full / 1e6, 4)}


out["reference_512_1thread"] = stock(512, 1, 1)
out["reference_512_allcores"] = stock(512, cores, cores)
out["port_512_allcores"] = port(512, 1)
out["reference_2048_1thread"] = stock(2048, 1, 1)
n = min(32, cores)                      # 1.2 GB of light vertices + grid per renderer: 32 renderers = 38 GB
out["reference_2048_%dthreads" % n] = stock(2048, n, n)
out["port_2048_allcores"] = port(2048, 8)
print(json.dumps(out, indent=1))
