#!/bin/bash
# r05s: the material table in LDS (against the build without it, equal stream priorities in both); helper streams at
# low priority on 2048^2 frames only -- the configurations of the full line again
set -u
export TMPDIR=/tmp
tag=${1:-r05s}
mkdir -p gpurun_out
out=gpurun_out/${tag}_ab.txt
: > $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_kat.py -x -q -m gpu > gpurun_out/${tag}_pytest.log 2>&1; grep -E "passed|failed|error" gpurun_out/${tag}_pytest.log | tail -3 | tee -a $out
B="python bench.py --no-cpu-baseline --no-configs --no-traffic --steps 12 --warmup 3"
p() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']['per_kernel']; print(d['value'], d['ms_per_step'], r['k_light_trace']['ms'], r['k_camera_trace']['ms'], r['k_connect_di+vc']['ms'], r['k_merge']['ms'])"; }
PREV=$PWD/smallvcm_amd/csrc/libsmallvcm_amd_prev.so
NEW=$PWD/smallvcm_amd/csrc/libsmallvcm_amd.so
export SMALLVCM_AMD_STREAM_PRIO=0
for rep in 1 2; do
  echo "== prev-$rep (no LDS materials)" | tee -a $out; SMALLVCM_AMD_LIB=$PREV $B 2>/dev/null | grep '^{' | p | tee -a $out
  echo "== new-$rep" | tee -a $out; SMALLVCM_AMD_LIB=$NEW $B 2>/dev/null | grep '^{' | p | tee -a $out
done
for v in prev new; do
  case $v in prev) L=$PREV;; new) L=$NEW;; esac
  echo "== $v-512" | tee -a $out; SMALLVCM_AMD_LIB=$L $B --res 512 2>/dev/null | grep '^{' | p | tee -a $out
  echo "== $v-s3" | tee -a $out; SMALLVCM_AMD_LIB=$L $B --scene 3 --res 1024 2>/dev/null | grep '^{' | p | tee -a $out
done
unset SMALLVCM_AMD_STREAM_PRIO
pc() { python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('headline', d['value'], d['ms_per_step'])
for c in d.get('configs',[]): print('  ', c['name'], c['value'], c['ms_per_step'], c['roofline'].get('grid_build_ms_side_stream'))"; }
echo "== new, defaults, with configs" | tee -a $out; python bench.py --no-cpu-baseline --no-traffic --steps 8 --warmup 2 2>/dev/null | grep '^{' | pc | tee -a $out
