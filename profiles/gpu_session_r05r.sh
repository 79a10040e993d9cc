#!/bin/bash
# r05r: why did the several-renderer configurations (and C1 inside the full line) fall?  stream priority / K3c stream
set -u
export TMPDIR=/tmp
tag=${1:-r05r}
mkdir -p gpurun_out
out=gpurun_out/${tag}_configs.txt
: > $out
B="python bench.py --no-cpu-baseline --no-traffic --steps 8 --warmup 2"
p() { python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('headline', d['value'], d['ms_per_step'])
for c in d.get('configs',[]): print('  ', c['name'], c['value'], c['ms_per_step'], c['roofline'].get('grid_build_ms_side_stream'))"; }
echo "== default" | tee -a $out; $B 2>/dev/null | grep '^{' | p | tee -a $out
echo "== prio0" | tee -a $out; SMALLVCM_AMD_STREAM_PRIO=0 $B 2>/dev/null | grep '^{' | p | tee -a $out
echo "== prio0 vc0" | tee -a $out; SMALLVCM_AMD_STREAM_PRIO=0 SMALLVCM_AMD_VC_STREAM=0 $B 2>/dev/null | grep '^{' | p | tee -a $out
echo "== vc0" | tee -a $out; SMALLVCM_AMD_VC_STREAM=0 $B 2>/dev/null | grep '^{' | p | tee -a $out
