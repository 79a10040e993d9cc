#!/bin/bash
# r05u: the query key's grid-header words and bucket geometry read once per kernel (they were two vector loads + a
# vmcnt(0) + a loop per appended camera vertex): parity subset, A/B against the previous commit, the bench line
set -u
export TMPDIR=/tmp
tag=${1:-r05u}
mkdir -p gpurun_out
out=gpurun_out/${tag}_ab.txt
: > $out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "full_frame or bit_exact or wavefront or sharded or bpm or ppm" > gpurun_out/${tag}_pytest.log 2>&1; grep -E "passed|failed|error" gpurun_out/${tag}_pytest.log | tail -3 | tee -a $out
B="python bench.py --no-cpu-baseline --no-configs --no-traffic --steps 12 --warmup 3"
p() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']['per_kernel']; print(d['value'], d['ms_per_step'], r['k_light_trace']['ms'], r['k_camera_trace']['ms'], r['k_connect_di+vc']['ms'], r['k_merge']['ms'])"; }
PREV=$PWD/smallvcm_amd/csrc/libsmallvcm_amd_prev.so
for rep in 1 2; do
  echo "== prev-$rep" | tee -a $out; SMALLVCM_AMD_LIB=$PREV $B 2>/dev/null | grep '^{' | p | tee -a $out
  echo "== new-$rep" | tee -a $out; $B 2>/dev/null | grep '^{' | p | tee -a $out
done
echo "== new-512" | tee -a $out; $B --res 512 2>/dev/null | grep '^{' | p | tee -a $out
echo "== new-bpm" | tee -a $out; $B --algo bpm 2>/dev/null | grep '^{' | p | tee -a $out
