#!/usr/bin/env python
"""Where the residence time of k_merge_pairs' waves goes (a variant built with -DVCM_K4_REGIONS; the clock is the wave's, in registers):
    SMALLVCM_AMD_LIB=profiles/ab_k4r/csrc/libsmallvcm_amd.so python profiles/tools/k4_regions.py [res] [iterations] [algo]
The figure is wave RESIDENCE time (issue + waiting + the other waves of the SIMD) summed over the waves; a region that ends with the use
of loaded data holds the wait for it."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from smallvcm_amd.renderer import VertexCM, cornell_scene, load_library
res = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
nit = int(sys.argv[2]) if len(sys.argv) > 2 else 12
algo = {"vcm": VertexCM.kVcm, "bpm": VertexCM.kBpm}[sys.argv[3] if len(sys.argv) > 3 else "vcm"]
L = load_library()
L.k4_regions_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
r = VertexCM(cornell_scene(1, res, res), algo, 0.003, 0.75, 1234)
r.mMaxPathLength = 10
for it in range(5):
    r.RunIteration(it)
r.framebuffer_sum()
assert L.k4_regions_read(None, 1) == 0
for it in range(5, 5 + nit):
    r.RunIteration(it)
r.framebuffer_sum()
buf = (C.c_ulonglong * 16)()
assert L.k4_regions_read(buf, 0) == 0
v = np.array(list(buf)[:8], dtype=np.float64)
names = ["set-up: the query, its row, its 8 cell ranges; the first two steps' loads issued",
         "scan: waiting for the step's candidates + the distances",
         "scan: the push (ballots, counts, ring writes)",
         "batch: ring read + the gathers issued",
         "batch: RangeQuery::Process (waits for the gathers of the batch before, the query rows)",
         "batch: the ordered rounds into the accumulators",
         "scan: the next step's position (run ends: LDS reads)",
         "end: the result written"]
print("k_merge_pairs, %d^2 %s, iterations 5..%d: wave residence time by region (100 MHz clock x waves)" % (res, sys.argv[3] if len(sys.argv) > 3 else "vcm", 4 + nit))
for n, x in zip(names, v):
    print("  %5.1f %%  %s" % (100.0 * x / v.sum(), n))
r.close()
