#!/bin/bash
# 512^2: kernel trace (gaps between launches), drop-in throughput through the unchanged driver
set -u
TAG=${1:-r02m}
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/${TAG}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats512 -- python bench.py --no-cpu-baseline --no-configs --no-traffic --res 512 --steps 20 --warmup 4 > ${O}_bench512.log 2>&1
python - ${TAG} <<'PY' | tee gpurun_out/${TAG}_trace512_summary.txt
import csv, glob, sys, collections
tag = sys.argv[1]
f = sorted(glob.glob("gpurun_out/%s_stats512/*/*kernel_trace.csv" % tag))[-1]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r.get("Stream_Id", r.get("Queue_Id", "0"))) for r in csv.DictReader(open(f))]
rows.sort()
# take the last 10 iterations: split at k_zero_ranges that follows a k_resolve
starts = [i for i, r in enumerate(rows) if "k_resolve" in r[2]]
lo, hi = starts[-11], starts[-1]
seg = rows[lo + 1:hi + 1]
t0, t1 = seg[0][0], seg[-1][1]
# union of busy intervals
busy, cur_s, cur_e = 0, None, None
for s, e, n, q in seg:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("10 iterations at 512^2: %.3f ms wall per iteration, GPU busy (union of kernel intervals) %.3f ms, idle %.3f ms, %d launches per iteration" % ((t1 - t0) / 1e7, busy / 1e7, (t1 - t0 - busy) / 1e7, len(seg) / 10))
per = collections.defaultdict(lambda: [0, 0])
for s, e, n, q in seg:
    per[n][0] += e - s; per[n][1] += 1
for n, (d, c) in sorted(per.items(), key=lambda kv: -kv[1][0])[:24]:
    print("  %-46s %7.1f us per iteration (%d launches)" % (n[:46], d / 1e4, c // 10))
PY
# the unchanged reference driver over the drop-in, one host thread: GPU iterations + per-iteration framebuffer refresh
for it in 64 256; do
  s=$(date +%s.%N)
  taskset -c 0 smallvcm_amd/dropin/smallvcm -s 1 -a vcm -i $it -o /tmp/dropin_$it.bmp > /tmp/dropin_$it.log 2>&1
  e=$(date +%s.%N)
  python -c "print('unchanged smallvcm over the drop-in, -s 1 -a vcm -i $it (512^2, one host thread): %.2f s wall = %.1f Mpaths/s incl. process start, scene build, per-iteration framebuffer refresh' % ($e - $s, 2 * 512 * 512 * $it / ($e - $s) / 1e6))"
done | tee ${O}_dropin.txt
python bench.py --no-cpu-baseline --no-configs --no-traffic --res 512 --steps 20 --warmup 4 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('512^2 bench: %.1f Mpaths/s %.3f ms' % (d['value'], d['ms_per_step']))" | tee -a ${O}_dropin.txt
python bench.py --no-cpu-baseline --no-configs --no-traffic --steps 12 --warmup 3 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('2048^2 bench: %.1f Mpaths/s %.3f ms' % (d['value'], d['ms_per_step']))" | tee -a ${O}_dropin.txt
