#!/bin/bash
# task sort for BVH scenes: the mesh / random / switch tests on the device, M1 against the previous revision
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_scene2.py tests/test_fuzz_scenes.py -q -m gpu -x 2>&1 | tail -3 > gpurun_out/r06ts_pytest.log
timeout 600 python -m pytest tests/test_gpu_switches.py -q -m gpu -x -k "BVH or default" 2>&1 | tail -3 >> gpurun_out/r06ts_pytest.log
ENVS="SMALLVCM_AMD_TASK_SORT=0" bash profiles/m1_base_ab.sh > gpurun_out/r06ts_m1.txt 2>&1
cat gpurun_out/r06ts_pytest.log gpurun_out/r06ts_m1.txt
