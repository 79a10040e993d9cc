#!/usr/bin/env python
"""Join the three runs of profiles/tools/fetch_calib (timing, --pmc FETCH_SIZE, --pmc WRITE_SIZE) into
profiles/<tag>_fetch_calib.json.  Usage (on the GPU box, from the repo root):
    python profiles/tools/fetch_calib.py <tag>          # runs the three passes itself, writes gpurun_out/<tag>_fetch_calib.json
Per pattern: useful bytes, bytes of the distinct 128-byte lines touched, what the counter reported, the factors
useful / reported and lines / reported, and the GB/s each reading implies for the un-profiled run."""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
EXE = os.path.join(ROOT, "profiles", "tools", "fetch_calib")


def run_plain():
    out = subprocess.run([EXE], capture_output=True, text=True, check=True).stdout
    rows = []
    for l in out.splitlines():
        if l.startswith("CALIB "):
            t = l.split()
            rows.append({"pattern": t[1], "useful_bytes": float(t[2].split("=")[1]), "line_bytes": float(t[3].split("=")[1]),
                         "ms": float(t[4].split("=")[1])})
    return rows


def run_pmc(counter, tag):
    d = os.path.join(ROOT, "gpurun_out", "%s_calib_%s" % (tag, counter.lower()))
    subprocess.run(["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--", EXE], cwd="/tmp",
                   env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    vals = []
    for r in sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"])):
        k = r["Kernel_Name"].replace("void ", "")
        if r["Counter_Name"] == counter and k.startswith(("read_", "write_", "atomic_")):
            vals.append(float(r["Counter_Value"]) * 1024.0)   # KB
    return [(vals[i] + vals[i + 1]) / 2 for i in range(0, len(vals) - 1, 2)]   # every pattern is launched twice


def main():
    tag = sys.argv[1]
    rows = run_plain()
    fetch, write = run_pmc("FETCH_SIZE", tag), run_pmc("WRITE_SIZE", tag)
    assert len(fetch) == len(rows) == len(write), (len(rows), len(fetch), len(write))
    for r, f, w in zip(rows, fetch, write):
        rep = w if r["pattern"].startswith("write_") else f
        r["FETCH_SIZE_bytes"], r["WRITE_SIZE_bytes"] = f, w
        r["useful_over_reported"] = round(r["useful_bytes"] / rep, 4) if rep else None
        r["lines_over_reported"] = round(r["line_bytes"] / rep, 4) if rep else None
        s = r["ms"] / 1e3
        r["GBs_if_useful"] = round(r["useful_bytes"] / s / 1e9, 1)
        r["GBs_if_lines"] = round(r["line_bytes"] / s / 1e9, 1)
        r["GBs_reported"] = round(rep / s / 1e9, 1)
        if r["pattern"].startswith("atomic_"):
            r["reported_fetch_plus_write_per_atomic"] = round((f + w) / (r["useful_bytes"] / 4), 2)
    json.dump({"note": "profiles/tools/fetch_calib.hip on one MI355X; counters = mean of the two dispatches of each pattern",
               "patterns": rows}, open(os.path.join(ROOT, "gpurun_out", "%s_fetch_calib.json" % tag), "w"), indent=1)
    for r in rows:
        print("%-28s useful %8.0f MB lines %8.0f MB  FETCH %8.0f MB WRITE %8.0f MB  useful/rep %-7s lines/rep %-7s %.3f ms  (%s / %s / %s GB/s)"
              % (r["pattern"], r["useful_bytes"] / 1e6, r["line_bytes"] / 1e6, r["FETCH_SIZE_bytes"] / 1e6, r["WRITE_SIZE_bytes"] / 1e6,
                 r["useful_over_reported"], r["lines_over_reported"], r["ms"], r["GBs_if_useful"], r["GBs_if_lines"], r["GBs_reported"]))


if __name__ == "__main__":
    main()
