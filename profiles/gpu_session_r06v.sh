#!/bin/bash
# vertex-indexed addends: parity on the device (the parity / switch / sharded suites), then the A/B against the previous revision
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_switches.py -q -m gpu -x 2>&1 | tail -4 > gpurun_out/r06v_pytest.log
ITER=40 WARM=5 BASE=1 REPS=4 bash profiles/quick_ab.sh > gpurun_out/r06v_ab.txt 2>&1
cat gpurun_out/r06v_pytest.log gpurun_out/r06v_ab.txt
