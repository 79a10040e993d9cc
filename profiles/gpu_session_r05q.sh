#!/bin/bash
# r05q: K3 reads its three queues' words together and the light path's length mask once per path; whole GPU suite
# the whole GPU suite on these sources; region clock with sample_scattering split up
set -u
export TMPDIR=/tmp
tag=${1:-r05q}
mkdir -p gpurun_out
out=gpurun_out/${tag}_ab.txt
: > $out
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/${tag}_pytest.log 2>&1; grep -E "passed|failed|error" gpurun_out/${tag}_pytest.log | tail -3 | tee -a $out
B="python bench.py --no-cpu-baseline --no-configs --no-traffic --steps 12 --warmup 3"
p() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']['per_kernel']; print(d['value'], d['ms_per_step'], r['k_light_trace']['ms'], r['k_camera_trace']['ms'], r['k_connect_di+vc']['ms'], r['k_merge']['ms'])"; }
PREV=$PWD/smallvcm_amd/csrc/libsmallvcm_amd_prev.so
for rep in 1 2; do
  echo "== prev-$rep" | tee -a $out; SMALLVCM_AMD_LIB=$PREV $B 2>/dev/null | grep '^{' | p | tee -a $out
  echo "== new-$rep" | tee -a $out; $B 2>/dev/null | grep '^{' | p | tee -a $out
done
for v in prev new; do
  case $v in prev) L=$PREV;; new) L=$PWD/smallvcm_amd/csrc/libsmallvcm_amd.so;; esac
  echo "== $v-512" | tee -a $out; SMALLVCM_AMD_LIB=$L $B --res 512 2>/dev/null | grep '^{' | p | tee -a $out
  echo "== $v-s3" | tee -a $out; SMALLVCM_AMD_LIB=$L $B --scene 3 --res 1024 2>/dev/null | grep '^{' | p | tee -a $out
done
export SMALLVCM_AMD_LIB=$PWD/smallvcm_amd/csrc/libsmallvcm_amd_rc.so
timeout 120 python profiles/tools/region_clock.py 1 2048 vcm > gpurun_out/${tag}_region_clock.txt 2>&1
cat gpurun_out/${tag}_region_clock.txt | head -50
