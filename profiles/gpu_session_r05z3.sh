#!/bin/bash
# round 3, final: A/B of the LDS scene tables against the build without them, the whole GPU suite, the full default bench
# line, the profile set and the region clock of the final sources (r05z)
set -u
TAG=r05z
export TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/${TAG}_lds_ab.txt
: > $out
B="python bench.py --no-cpu-baseline --no-configs --no-traffic --steps 12 --warmup 3"
p() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']['per_kernel']; print(d['value'], d['ms_per_step'], r['k_light_trace']['ms'], r['k_camera_trace']['ms'], r['k_connect_di+vc']['ms'], r['k_merge']['ms'])"; }
PREV=$PWD/smallvcm_amd/csrc/libsmallvcm_amd_prev.so
for rep in 1 2; do
  echo "== prev-$rep (no LDS tables)" | tee -a $out; SMALLVCM_AMD_LIB=$PREV $B 2>/dev/null | grep '^{' | p | tee -a $out
  echo "== new-$rep" | tee -a $out; $B 2>/dev/null | grep '^{' | p | tee -a $out
done
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/${TAG}_pytest.log 2>&1 < /dev/null; grep -E "passed|failed" gpurun_out/${TAG}_pytest.log | tail -2
timeout 1500 python bench.py > gpurun_out/${TAG}_bench_default.log 2> gpurun_out/${TAG}_bench_default.err; tail -c 600 gpurun_out/${TAG}_bench_default.log
BENCH_ARGS="" timeout 900 bash profiles/collect.sh ${TAG} > gpurun_out/${TAG}_collect.log 2>&1; tail -2 gpurun_out/${TAG}_collect.log
timeout 300 python bench.py --gpus 2 --collectives threads --steps 8 --warmup 2 > gpurun_out/${TAG}_bench_2ranks_threads.log 2>&1; tail -c 300 gpurun_out/${TAG}_bench_2ranks_threads.log
SMALLVCM_AMD_LIB=$PWD/smallvcm_amd/csrc/libsmallvcm_amd_rc.so timeout 120 python profiles/tools/region_clock.py 1 2048 vcm > gpurun_out/${TAG}_region_clock.txt 2>&1
SMALLVCM_AMD_LIB=$PWD/smallvcm_amd/csrc/libsmallvcm_amd_rc.so timeout 120 python profiles/tools/region_clock.py 1 512 vcm >> gpurun_out/${TAG}_region_clock.txt 2>&1
