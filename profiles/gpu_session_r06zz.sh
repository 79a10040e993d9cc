#!/bin/bash
# last kernel change of round 4 (arrival store deferred by a bounce): a parity subset, then the profile set's counter passes
# and kernel stats at these sources (the GPU suite and the full bench line of r06z stay: 265 passed / 1023 Mpaths/s one commit earlier)
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "hip_equals_oracle or late_iterations" 2>&1 | tail -2 > gpurun_out/r06zz_pytest.log
BENCH_ARGS="--steps 20 --warmup 5" timeout 400 bash profiles/collect.sh r06z > gpurun_out/r06z_collect.log 2>&1
cat gpurun_out/r06zz_pytest.log; tail -1 gpurun_out/r06z_bench.log | cut -c1-200
