#!/usr/bin/env python
"""One VCM iteration of scene 1 at 4096^2 (16.8 M light + 16.8 M camera paths, four times BASELINE's largest frame) on the
device against the oracle: tape, counters, framebuffer bit for bit.  Index arithmetic beyond 2^31 bytes / floats per array is what
this is for (light store 10.7 GB, 2.2 G floats of merge records).   python profiles/tools/big_frame_check.py [res]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_lib import Oracle  # noqa: E402
from smallvcm_amd.renderer import VertexCM, cornell_scene  # noqa: E402

res = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sc = cornell_scene(1, res, res)
t = time.time()
r = VertexCM(sc, 4, 0.003, 0.75, 1234)
r.mMaxPathLength = 10
r.RunIteration(0)
r.backend.synchronize()
lc, cc = r.backend.rng_counts()
sg = r.stats()
fb = r.framebuffer_sum()
tg = time.time() - t
r.close()
t = time.time()
o = Oracle(sc, 4, threads=os.cpu_count() or 1)
o.run_iteration(0, 0, 10)
olc, occ = o.counts()
so = o.stats()
to = time.time() - t
ok = bool(np.array_equal(lc, olc) and np.array_equal(cc, occ))
keys = ("lightVertices", "lightRays", "cameraRays", "shadowRays", "mergeQueries", "mergeCandidates", "mergeAccepted", "connections", "lightSplats")
okc = all(so[k] == sg[k] for k in keys)
okf = bool(np.array_equal(fb.view(np.uint32), o.framebuffer().view(np.uint32)))
print("%dx%d VCM iteration 0: tape %s, counters %s, framebuffer %s  (device incl. set-up %.1f s, oracle %.1f s); %s" %
      (res, res, "equal" if ok else "DIFFERS", "equal" if okc else "DIFFER", "bit-identical" if okf else "DIFFERS", tg, to,
       {k: int(sg[k]) for k in ("lightVertices", "mergeCandidates", "connections")}), flush=True)
sys.exit(0 if (ok and okc and okf) else 1)
