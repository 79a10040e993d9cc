#!/bin/bash
# round 3, final set: the full default bench line (configs, live counters, CPU legs) and the rocprofv3 profile set of
# the same sources (kernel stats, FETCH / WRITE, SQ passes): profiles/r05z_*
set -u
TAG=${1:-r05z}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python bench.py > gpurun_out/${TAG}_bench_default.log 2> gpurun_out/${TAG}_bench_default.err; tail -c 1500 gpurun_out/${TAG}_bench_default.log
BENCH_ARGS="" timeout 900 bash profiles/collect.sh ${TAG} > gpurun_out/${TAG}_collect.log 2>&1; tail -3 gpurun_out/${TAG}_collect.log
timeout 300 python bench.py --gpus 2 --collectives threads --steps 8 --warmup 2 > gpurun_out/${TAG}_bench_2ranks_threads.log 2>&1; tail -c 800 gpurun_out/${TAG}_bench_2ranks_threads.log
