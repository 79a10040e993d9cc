#!/usr/bin/env python
"""How full the chip is over the life of ONE launch of k_merge_pairs (a variant built with -DVCM_K4_TIMES records when every workgroup
started and ended):   SMALLVCM_AMD_LIB=profiles/ab_k4t/csrc/libsmallvcm_amd.so python profiles/tools/k4_tail.py [res] [iterations]
prints the number of workgroups in flight at ten points of the launch, the time the last 10 % / 1 % of the workgroups end, and the
launch's length against the mean occupancy (what a perfectly packed launch would take)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from smallvcm_amd.renderer import VertexCM, cornell_scene, load_library
res = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
nit = int(sys.argv[2]) if len(sys.argv) > 2 else 6
r = VertexCM(cornell_scene(1, res, res), VertexCM.kVcm, 0.003, 0.75, 1234)
r.mMaxPathLength = 10
for it in range(nit):
    r.RunIteration(it)
r.framebuffer_sum()
L = load_library()
buf = (C.c_ulonglong * (2 * 32768))()
assert L.k4_times_read(buf) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 2).astype(np.float64)
t = t[t[:, 1] > 0]
t0 = t[:, 0].min()
s, e = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0   # 100 MHz clock -> us
total = e.max()
print("%d workgroups, launch %.0f us; a workgroup lives %.0f us on average, the longest %.0f" % (len(t), total, (e - s).mean(), (e - s).max()))
for f in (0.1, 0.3, 0.5, 0.7, 0.8, 0.9, 0.95, 0.98, 0.995):
    x = f * total
    print("  at %4.1f %% of the launch (%6.0f us): %5d workgroups in flight" % (100 * f, x, int(((s <= x) & (e > x)).sum())))
area = (e - s).sum()
peak = max(int(((s <= x) & (e > x)).sum()) for x in np.linspace(0, total, 200))
print("workgroup-time %.0f us x workgroups; at the peak occupancy of %d the same work packs into %.0f us (%.1f %% of the launch)" % (area, peak, area / peak, 100 * area / peak / total))
r.close()
