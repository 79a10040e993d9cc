#!/bin/bash
# r05t: K3's queue words through an LDS pointer (they were flat loads / stores with vmcnt(0) each): parity, A/B against
# the sources of the r05z set, region clock, the bench line with its live counters (no CPU legs)
set -u
export TMPDIR=/tmp
tag=${1:-r05t}
mkdir -p gpurun_out
out=gpurun_out/${tag}_ab.txt
: > $out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/${tag}_pytest.log 2>&1; grep -E "passed|failed|error" gpurun_out/${tag}_pytest.log | tail -3 | tee -a $out
B="python bench.py --no-cpu-baseline --no-configs --no-traffic --steps 12 --warmup 3"
p() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']['per_kernel']; print(d['value'], d['ms_per_step'], r['k_light_trace']['ms'], r['k_camera_trace']['ms'], r['k_connect_di+vc']['ms'], r['k_merge']['ms'])"; }
PREV=$PWD/smallvcm_amd/csrc/libsmallvcm_amd_prev.so
for rep in 1 2; do
  echo "== prev-$rep (r05z sources)" | tee -a $out; SMALLVCM_AMD_LIB=$PREV $B 2>/dev/null | grep '^{' | p | tee -a $out
  echo "== new-$rep" | tee -a $out; $B 2>/dev/null | grep '^{' | p | tee -a $out
done
SMALLVCM_AMD_LIB=$PWD/smallvcm_amd/csrc/libsmallvcm_amd_rc.so timeout 120 python profiles/tools/region_clock.py 1 2048 vcm > gpurun_out/${tag}_region_clock.txt 2>&1
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/${tag}_bench_nocpu.log 2> gpurun_out/${tag}_bench_nocpu.err
python - <<'PY' | tee -a $out
import json
for line in open('gpurun_out/r05t_bench_nocpu.log'):
    if line.startswith('{'):
        d=json.loads(line); print('headline', d['value'], d['ms_per_step'])
        for c in d.get('configs',[]): print('  ', c['name'], c['value'], c['ms_per_step'])
PY
