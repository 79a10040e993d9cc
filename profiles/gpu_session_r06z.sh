#!/bin/bash
# round 4, final set on the final sources: the GPU suite, the full default bench line as the driver runs it (configs, live
# counters on the run's own window, CPU legs), the rocprofv3 profile set of the same command (kernel stats, FETCH / WRITE, SQ
# passes): profiles/r06z_*
set -u
TAG=${1:-r06z}
export TMPDIR=/tmp
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; tail -2 gpurun_out/${TAG}_pytest.log
timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_default.log 2> gpurun_out/${TAG}_bench_default.err; tail -c 600 gpurun_out/${TAG}_bench_default.log
BENCH_ARGS="--steps 20 --warmup 5" timeout 900 bash profiles/collect.sh ${TAG} > gpurun_out/${TAG}_collect.log 2>&1; tail -3 gpurun_out/${TAG}_collect.log
