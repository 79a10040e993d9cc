#!/usr/bin/env python
"""Condense the raw rocprofv3 output of profiles/collect.sh (gpurun_out/<tag>_*) into the files kept under profiles/:
  <tag>_kernel_stats.csv   the --kernel-trace --stats table as rocprofv3 wrote it
  <tag>_traffic.json       per kernel and launch: HBM bytes = 2 x FETCH_SIZE (KB; the gfx950 correction of
                           MI355X_MICROARCH.md, section HBM) and WRITE_SIZE (KB), mean over the dispatches after the
                           first two (allocation / first-touch effects); what bench.py reports as roofline.traffic
  <tag>_pmc_sq.json        SQ counters per kernel (mean per dispatch) from the two SQ passes
Usage: python profiles/summarize.py <tag>"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out")
dst = os.path.join(ROOT, "profiles")


def one(pattern):
    f = sorted(glob.glob(os.path.join(src, pattern)))
    if not f:
        raise SystemExit("missing " + pattern)
    return f[-1]


def short(name):
    return name.split("(")[0].replace("void ", "")


def per_kernel(path, skip=2):
    """{kernel: {counter: mean value per dispatch}}, and mean duration in us"""
    vals, seen, dur = collections.defaultdict(lambda: collections.defaultdict(list)), collections.Counter(), collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        key = (k, r["Counter_Name"])
        seen[key] += 1
        if seen[key] <= skip:
            continue
        vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return ({k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in vals.items()},
            {k: sum(v) / len(v) for k, v in dur.items()})


shutil.copy(one("%s_stats/*/*kernel_stats.csv" % tag), os.path.join(dst, "%s_kernel_stats.csv" % tag))
fetch, dur = per_kernel(one("%s_fetch/*/*counter_collection.csv" % tag))
write, _ = per_kernel(one("%s_write/*/*counter_collection.csv" % tag))
kernels = {}
for k in sorted(fetch, key=lambda k: -dur.get(k, 0)):
    if not k.startswith("vcm::"):
        continue
    kernels[k] = {"avg_us": round(dur[k], 1), "fetch_bytes_x2": int(2 * 1024 * fetch[k].get("FETCH_SIZE", 0)),
                  "write_bytes": int(1024 * write.get(k, {}).get("WRITE_SIZE", 0))}
sys.path.insert(0, ROOT)
from bench import kernel_source_hash  # noqa: E402  (the sources the counters were collected on: run this BEFORE editing kernels)
json.dump({"kernel_src_sha16": kernel_source_hash(),
           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py scene 1 vcm 2048^2, mean over dispatches "
                   "after the first two; FETCH_SIZE (KB) doubled per MI355X_MICROARCH.md section HBM; WRITE_SIZE (KB) uncalibrated; "
                   "avg_us is the duration under counter collection (serialised dispatches)",
           "kernels": kernels}, open(os.path.join(dst, "%s_traffic.json" % tag), "w"), indent=1)
sq = {}
for p in ("sq1", "sq2", "sq3", "sq4"):
    if p in ("sq3", "sq4") and not glob.glob(os.path.join(src, "%s_%s/*/*counter_collection.csv" % (tag, p))):
        continue
    c, _ = per_kernel(one("%s_%s/*/*counter_collection.csv" % (tag, p)))
    for k, d in c.items():
        if k.startswith("vcm::"):
            sq.setdefault(k, {}).update({n: int(v) for n, v in d.items()})
json.dump({"note": "rocprofv3 --pmc SQ_* (one pass per group of counters), mean per dispatch after the first two", "kernels": sq},
          open(os.path.join(dst, "%s_pmc_sq.json" % tag), "w"), indent=1)
print("wrote profiles/%s_kernel_stats.csv, %s_traffic.json, %s_pmc_sq.json" % (tag, tag, tag))
