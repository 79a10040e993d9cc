#!/bin/bash
# round 3, final: the whole GPU suite, the full default bench line and the profile set of the final sources (r05z)
set -u
TAG=r05z
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/${TAG}_pytest.log 2>&1 < /dev/null; grep -E "passed|failed" gpurun_out/${TAG}_pytest.log | tail -2
timeout 1500 python bench.py > gpurun_out/${TAG}_bench_default.log 2> gpurun_out/${TAG}_bench_default.err; tail -c 600 gpurun_out/${TAG}_bench_default.log
BENCH_ARGS="" timeout 900 bash profiles/collect.sh ${TAG} > gpurun_out/${TAG}_collect.log 2>&1; tail -2 gpurun_out/${TAG}_collect.log
timeout 300 python bench.py --gpus 2 --collectives threads --steps 8 --warmup 2 > gpurun_out/${TAG}_bench_2ranks_threads.log 2>&1; tail -c 300 gpurun_out/${TAG}_bench_2ranks_threads.log
