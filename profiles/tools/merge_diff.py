#!/usr/bin/env python
"""How far the frame of one range-merge kernel is from another's (GPU box):
    python profiles/tools/merge_diff.py [kindA=walk] [kindB=pairs] [res=256] [iterations=3] [scene=1] [algo=4]
prints the number of texels that differ, the RMSE and the largest relative difference; counters of both."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from smallvcm_amd.renderer import VertexCM, cornell_scene
a = sys.argv[1] if len(sys.argv) > 1 else "walk"
b = sys.argv[2] if len(sys.argv) > 2 else "pairs"
res = int(sys.argv[3]) if len(sys.argv) > 3 else 256
nit = int(sys.argv[4]) if len(sys.argv) > 4 else 3
sid = int(sys.argv[5]) if len(sys.argv) > 5 else 1
algo = int(sys.argv[6]) if len(sys.argv) > 6 else 4
fr = {}
for kind in (a, b):
    r = VertexCM(cornell_scene(sid, res, res), algo, 0.003, 0.75, 1234)
    r.backend.set_merge_kernel(kind)
    r.mMaxPathLength = 10
    for it in range(nit):
        r.RunIteration(it)
    fr[kind] = r.framebuffer_sum().astype(np.float64)
    print(kind, {k: v for k, v in r.stats().items() if k.startswith("merge")})
    r.close()
d = fr[a] - fr[b]
n = int(np.count_nonzero(d))
rel = np.abs(d) / np.maximum(np.abs(fr[a]), 1e-30)
print("texel components that differ: %d of %d; RMSE %.3e (mean %.4f); max |rel| %.3e; max |abs| %.3e"
      % (n, d.size, float(np.sqrt(np.mean(d * d))), float(fr[a].mean()), float(rel.max()), float(np.abs(d).max())))
