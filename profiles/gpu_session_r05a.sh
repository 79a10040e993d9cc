#!/bin/bash
# round 3, first call: the new host paths (farm C-ABI, bench --gpus N as typed), ADVICE fixes, FETCH_SIZE calibration,
# the bench line with live VALU / traffic counters
set -u
TAG=${1:-r05a}
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/${TAG}
timeout 900 python -m pytest tests/test_gpu_dropin_sharded.py tests/test_gpu_parity.py -q -m gpu -x \
   -k "farm or bench_line or local_bbox or end_iteration or cpp_host_equals or (hip_equals_oracle and wavefront and 64) or render_farm_on_one_gpu or rccl_plumbing" \
   > ${O}_pytest.log 2>&1 < /dev/null; tail -3 ${O}_pytest.log
( cd /tmp && timeout 600 python /root/repo/profiles/tools/fetch_calib.py ${TAG} ) > ${O}_calib.log 2>&1; tail -30 ${O}_calib.log
timeout 600 python bench.py --cpu-baseline port --no-configs --steps 12 --warmup 3 > ${O}_bench.log 2> ${O}_bench.err; tail -c 6000 ${O}_bench.log; tail -5 ${O}_bench.err
