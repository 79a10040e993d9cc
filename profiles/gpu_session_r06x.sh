#!/bin/bash
# 48-byte slot records: parity on the device, the A/B against the previous revision, fetch / write of the kernels that touch them
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "baseline or equals_oracle or late" 2>&1 | tail -3 > gpurun_out/r06x_pytest.log
ITER=40 WARM=5 BASE=1 REPS=4 bash profiles/quick_ab.sh > gpurun_out/r06x_ab.txt 2>&1
KERNELS="k_resolve k_connect_di k_camera_trace k_merge_walk" bash profiles/fetch_ab.sh > gpurun_out/r06x_fetch.txt 2>&1
cat gpurun_out/r06x_pytest.log gpurun_out/r06x_ab.txt gpurun_out/r06x_fetch.txt
