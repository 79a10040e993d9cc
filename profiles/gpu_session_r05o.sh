#!/bin/bash
# r05o: rectangle loop unrolled by two (wait, then issue the next entry's loads), min3 / max3 classification variant
set -u
export TMPDIR=/tmp
tag=${1:-r05o}
mkdir -p gpurun_out
out=gpurun_out/${tag}_ab.txt
: > $out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3 | tee -a $out
B="python bench.py --no-cpu-baseline --no-configs --no-traffic --steps 12 --warmup 3"
p() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']['per_kernel']; print(d['value'], d['ms_per_step'], r['k_light_trace']['ms'], r['k_camera_trace']['ms'], r['k_connect_di+vc']['ms'], r['k_merge']['ms'])"; }
PREV=$PWD/smallvcm_amd/csrc/libsmallvcm_amd_prev.so
MM=$PWD/smallvcm_amd/csrc/libsmallvcm_amd_mm.so
for rep in 1 2; do
  echo "== prev-$rep" | tee -a $out; SMALLVCM_AMD_LIB=$PREV $B 2>/dev/null | grep '^{' | p | tee -a $out
  echo "== new-$rep" | tee -a $out; $B 2>/dev/null | grep '^{' | p | tee -a $out
  echo "== mm-$rep" | tee -a $out; SMALLVCM_AMD_LIB=$MM $B 2>/dev/null | grep '^{' | p | tee -a $out
done
for v in prev new mm; do
  case $v in prev) L=$PREV;; mm) L=$MM;; new) L=$PWD/smallvcm_amd/csrc/libsmallvcm_amd.so;; esac
  echo "== $v-512" | tee -a $out; SMALLVCM_AMD_LIB=$L $B --res 512 2>/dev/null | grep '^{' | p | tee -a $out
  echo "== $v-s3" | tee -a $out; SMALLVCM_AMD_LIB=$L $B --scene 3 --res 1024 2>/dev/null | grep '^{' | p | tee -a $out
done
SMALLVCM_AMD_LIB=$MM timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "full_frame or long_paths or bit_exact" 2>&1 | tail -2 | tee -a $out
export SMALLVCM_AMD_LIB=$PWD/smallvcm_amd/csrc/libsmallvcm_amd_rc.so
timeout 120 python profiles/tools/region_clock.py 1 2048 vcm > gpurun_out/${tag}_region_clock.txt 2>&1
cat gpurun_out/${tag}_region_clock.txt | head -32
