#!/usr/bin/env python
"""Per-iteration launch trace of the bench loop at a given resolution (run on the GPU box, from the repo root):
    python profiles/tools/trace_summary.py <tag> [--res 512] [--scene 1] [--algo vcm]
rocprofv3 --kernel-trace around `bench.py --child` (10 iterations after 2 warm-up), then per kernel: time per iteration
and launches per iteration; wall time per iteration, the union of the kernel intervals (GPU busy) and the idle rest.
-> gpurun_out/<tag>_trace<res>_summary.txt"""
import collections
import csv
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag = sys.argv[1]
args = sys.argv[2:]
res = args[args.index("--res") + 1] if "--res" in args else "512"
steps, warm = 10, 2
d = os.path.join(ROOT, "gpurun_out", "%s_trace%s" % (tag, res))
cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.join(ROOT, "bench.py"), "--child",
       "--steps", str(steps), "--warmup", str(warm), "--res", res] + [a for a in args if a not in ("--res", res)]
subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the timed iterations = the last `steps` launches of k_resolve delimit them
ends = [int(r["End_Timestamp"]) for r in rows if "k_resolve" in r["Kernel_Name"]]
t0, t1 = ends[-steps - 1], ends[-1]
sel = [r for r in rows if int(r["Start_Timestamp"]) >= t0 and int(r["End_Timestamp"]) <= t1]
per, cnt = collections.Counter(), collections.Counter()
iv = []
for r in sel:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    per[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); cnt[k] += 1
    iv.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
iv.sort()
busy, cur_s, cur_e = 0, None, None
for s, e in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
if cur_e is not None: busy += cur_e - cur_s
out = ["%d iterations at %s^2: %.3f ms wall per iteration, GPU busy (union of kernel intervals) %.3f ms, idle %.3f ms, %.1f launches per iteration"
       % (steps, res, (t1 - t0) / steps / 1e6, busy / steps / 1e6, (t1 - t0 - busy) / steps / 1e6, len(sel) / steps)]
for k, v in per.most_common():
    out.append("  %-48s %7.1f us per iteration (%.1f launches)" % (k[:48], v / steps / 1e3, cnt[k] / steps))
txt = "\n".join(out)
open(os.path.join(ROOT, "gpurun_out", "%s_trace%s_summary.txt" % (tag, res)), "w").write(txt + "\n")
print(txt)
