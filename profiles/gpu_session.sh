#!/bin/bash
# One GPU-box session (run through gpurun from the repo root): parity tests, A/B of the merge kernels,
# the full bench line, rocprofv3 kernel stats.  Everything lands in gpurun_out/<tag>_*.
#   gpurun --timeout 1500 -- 'bash profiles/gpu_session.sh r02a'
set -u
TAG=${1:-r02x}
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/${TAG}
( rocm-smi --showproductname 2>/dev/null | head -8; nproc; free -g | head -2 ) > ${O}_box.log 2>&1
# 1. parity (the new kernels first)
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu --maxfail=8 -x -k "hip_equals_oracle or long_paths or other_box" > ${O}_pytest_parity.log 2>&1
echo "parity rc=$?" >> ${O}_box.log
tail -3 ${O}_pytest_parity.log
# 2. A/B of the merge kernels, same box, same clocks
for m in lane staged lane staged; do
  SMALLVCM_AMD_MERGE=$m timeout 300 python bench.py --no-cpu-baseline --no-configs --no-traffic --steps 12 --warmup 3 >> ${O}_ab_$m.log 2>&1
done
grep -h -o '"value": [0-9.]*\|"k_merge": {"ms": [0-9.]*' ${O}_ab_lane.log ${O}_ab_staged.log
# 3. the whole suite
timeout 1200 python -m pytest tests -q -m gpu --maxfail=8 > ${O}_pytest_all.log 2>&1
echo "all rc=$?" >> ${O}_box.log
tail -3 ${O}_pytest_all.log
# 4. full bench line (configs, live traffic, port baseline) and the reference leg on a small frame
timeout 900 python bench.py --cpu-baseline port > ${O}_bench.log 2>&1
timeout 300 python bench.py --res 512 --no-traffic --cpu-baseline reference > ${O}_bench512_ref.log 2>&1
tail -c 600 ${O}_bench.log
# 5. rocprofv3 kernel stats of the headline run
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats -- python bench.py --no-cpu-baseline --no-configs --no-traffic > ${O}_prof_bench.log 2>&1
ls gpurun_out/${TAG}_stats/*/ 2>/dev/null | head
