#!/bin/bash
# kernel stats of the v2-scene iterations (every command under its own timeout)
set -u
TAG=${1:-r02w}
export TMPDIR=/tmp
mkdir -p gpurun_out
for w in tilted bumpy; do
  timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_$w -- python profiles/tools/mesh_prof.py $w > gpurun_out/${TAG}_$w.log 2>&1 < /dev/null
  f=$(ls gpurun_out/${TAG}_$w/*/*kernel_stats.csv 2>/dev/null | tail -1)
  echo "== $w"
  if [ -n "$f" ]; then head -14 "$f" | cut -d, -f1-4 | cut -c1-150; else tail -3 gpurun_out/${TAG}_$w.log; fi
done
