#!/usr/bin/env python
"""One iteration's launches as a timeline (run on the GPU box, from the repo root):
    python profiles/tools/timeline.py <tag> [--res 2048] [--scene 1] [--algo vcm]
rocprofv3 --kernel-trace around `bench.py --child` (6 iterations after 2 warm-up); prints, for the LAST iteration, every
dispatch with its queue, start offset, duration -- which kernels overlap, where the main stream waits.
-> gpurun_out/<tag>_timeline<res>.txt"""
import csv
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag = sys.argv[1]
args = sys.argv[2:]
res = args[args.index("--res") + 1] if "--res" in args else "2048"
d = os.path.join(ROOT, "gpurun_out", "%s_tl%s" % (tag, res))
cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.join(ROOT, "bench.py"), "--child",
       "--steps", "6", "--warmup", "2", "--res", res] + [a for a in args if a not in ("--res", res)]
subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ends = [int(r["End_Timestamp"]) for r in rows if "k_resolve" in r["Kernel_Name"]]
t0, t1 = ends[-2], ends[-1]
sel = [r for r in rows if int(r["Start_Timestamp"]) >= t0 - 200000 and int(r["End_Timestamp"]) <= t1]
qkey = "Queue_Id" if "Queue_Id" in rows[0] else ("Stream_Id" if "Stream_Id" in rows[0] else None)
out = ["last iteration: %.3f ms between the ends of two k_resolve; columns: start offset [us], duration [us], queue, kernel" % ((t1 - t0) / 1e6)]
for r in sel:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("vcm::", "")
    out.append("%9.1f %8.1f  q%-3s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                        r.get(qkey, "?") if qkey else "?", k[:70]))
txt = "\n".join(out)
open(os.path.join(ROOT, "gpurun_out", "%s_timeline%s.txt" % (tag, res)), "w").write(txt + "\n")
print(txt)
