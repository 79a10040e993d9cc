#!/usr/bin/env python
"""Per-kernel means of the rocprofv3 --pmc passes of profiles/gpu_session_pmc.sh:
    python profiles/tools/pmc_table.py gpurun_out/r02e [out.json]
prints one block per kernel (mean per dispatch after the first 2) and optionally writes them as JSON."""
import collections, csv, glob, json, sys
base = sys.argv[1]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in sorted(glob.glob(base + "_pmc*/*/*counter_collection.csv")):
    seen = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not k.startswith("vcm::"):
            continue
        seen[(k, r["Counter_Name"])] += 1
        if seen[(k, r["Counter_Name"])] <= 2:
            continue
        vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {}
for k in sorted(vals, key=lambda k: -sum(dur[k]) / len(dur[k])):
    d = {c: sum(v) / len(v) for c, v in vals[k].items()}
    d["avg_us_profiled"] = sum(dur[k]) / len(dur[k])
    out[k] = d
want = sys.argv[3:] if len(sys.argv) > 3 else None
for k, d in out.items():
    if d["avg_us_profiled"] < 150:
        continue
    print("==", k, "%.0f us" % d["avg_us_profiled"])
    wc = d.get("SQ_WAVE_CYCLES", 0)
    for c in sorted(d):
        if c == "avg_us_profiled":
            continue
        extra = ""
        if wc and c.startswith(("SQ_WAIT", "SQ_ACTIVE", "SQ_INST_CYCLES", "SQ_BUSY")):
            extra = "  (%.1f %% of wave cycles)" % (100 * d[c] / wc)
        print("   %-30s %16.0f%s" % (c, d[c], extra))
if len(sys.argv) > 2 and sys.argv[2] != "-":
    json.dump({"note": "rocprofv3 --pmc passes (profiles/gpu_session_pmc.sh), mean per dispatch after the first two", "kernels": out},
              open(sys.argv[2], "w"), indent=1)
