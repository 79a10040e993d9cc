#!/usr/bin/env python
"""How many of a wave's 64 lanes have candidates left, step by step, in k_merge_pairs -- and what other ways of dealing the queries to
lanes would make of it.  A variant built with -DVCM_K4_STEPS records the scan steps (blocks of four candidates over its runs) of
every query in the order the kernel takes them:
    SMALLVCM_AMD_LIB=profiles/ab_k4s/csrc/libsmallvcm_amd.so python profiles/tools/k4_lanes.py [res] [iterations] [algo]
Model: a wave steps until its slowest lane is done (the kernel); `refill T`: a lane that is done takes the wave's next query as soon as
at least T lanes are done (queries in the same order, a wave works through a contiguous range of them)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from smallvcm_amd.renderer import VertexCM, cornell_scene, load_library
res = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
nit = int(sys.argv[2]) if len(sys.argv) > 2 else 6
algo = {"vcm": VertexCM.kVcm, "bpm": VertexCM.kBpm}[sys.argv[3] if len(sys.argv) > 3 else "vcm"]
r = VertexCM(cornell_scene(1, res, res), algo, 0.003, 0.75, 1234)
r.mMaxPathLength = 10
for it in range(nit):
    r.RunIteration(it)
r.framebuffer_sum()
nq = int(r.backend.stats_at(0)["mergeQueries"])
print("mergeQueries of the last iteration:", nq)
L = load_library()
N = 1 << 24
buf = (C.c_ushort * N)()
assert L.k4_steps_read(buf, N) == 0
s = np.frombuffer(buf, dtype=np.uint16).astype(np.int64)
if nq is None:
    nq = int(np.nonzero(s)[0].max()) + 1
s = s[:nq]
print("%d queries, %.1f steps each on average (max %d); %d with no candidates" % (nq, s.mean(), s.max(), int((s == 0).sum())))
pad = (-nq) % 64
w = np.concatenate([s, np.zeros(pad, np.int64)]).reshape(-1, 64)
wave_steps = w.max(axis=1)
print("kernel's dealing: %d wave-steps, lanes with work %.3f" % (wave_steps.sum(), s.sum() / (64.0 * wave_steps.sum())))
# the same with the 256 queries of a workgroup dealt to its four waves by step count
pad = (-nq) % 256
g = np.sort(np.concatenate([s, np.zeros(pad, np.int64)]).reshape(-1, 256), axis=1).reshape(-1, 64)
print("a workgroup's 256 sorted by steps: %d wave-steps, lanes with work %.3f" % (g.max(axis=1).sum(), s.sum() / (64.0 * g.max(axis=1).sum())))
def refill(steps, span, T):
    """waves work through `span` consecutive queries each; returns wave-steps"""
    total = 0
    for a in range(0, len(steps), span):
        q = steps[a:a + span]
        nxt = min(64, len(q))
        left = np.zeros(64, np.int64)
        left[:nxt] = q[:nxt]
        while True:
            busy = left > 0
            nb = int(busy.sum())
            if nxt < len(q) and 64 - nb >= T:
                idle = np.nonzero(~busy)[0]
                k = min(len(idle), len(q) - nxt)
                left[idle[:k]] = q[nxt:nxt + k]
                nxt += k
                continue
            if nb == 0:
                break
            # advance to the next event: the step at which enough lanes are done, or all are
            v = np.sort(left[busy])
            if nxt < len(q):
                need = max(T - (64 - nb), 1)
                d = int(v[min(need, len(v)) - 1])
            else:
                d = int(v[-1])
            total += d
            left = np.maximum(left - d, 0)
    return total
sample = s[: min(nq, 64 * 20000)]
base = np.concatenate([sample, np.zeros((-len(sample)) % 64, np.int64)]).reshape(-1, 64).max(axis=1).sum()
print("sample of %d queries: kernel's dealing %d wave-steps" % (len(sample), base))
for span in (256, 1024, 4096):
    for T in (1, 8, 16, 32):
        t = refill(sample, span, T)
        print("  refill at >= %2d idle lanes, %4d queries per wave: %d wave-steps (%.3f of the kernel's), lanes with work %.3f"
              % (T, span, t, t / float(base), sample.sum() / (64.0 * t)))
r.close()
