#!/usr/bin/env python
"""Mpaths/s of the procedural mesh room (tests/mesh_scenes.py) at a given tessellation: VCM, 1024^2 by default.
    python profiles/tools/mesh_rate.py <grid> [res] [iterations]      (environment switches apply: SMALLVCM_AMD_TASK_SORT=0 ...)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from mesh_scenes import bumpy_room  # noqa: E402
from smallvcm_amd.renderer import VertexCM  # noqa: E402

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 72
res = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
sc = bumpy_room(grid=grid, resx=res, resy=res)
r = VertexCM(sc, 4, 0.003, 0.75, 1234)
r.mMaxPathLength = 10
for it in range(4):
    r.RunIteration(it)
r.backend.synchronize()
t = time.time()
for it in range(4, 4 + n):
    r.RunIteration(it)
r.backend.synchronize()
dt = (time.time() - t) / n
print("grid %d: %d primitives, %dx%d, TASK_SORT=%s: %.3f ms per iteration, %.1f Mpaths/s" % (grid, sc.nPrims, res, res, os.environ.get("SMALLVCM_AMD_TASK_SORT", "default"), dt * 1e3, 2 * res * res / dt / 1e6), flush=True)
r.close()
