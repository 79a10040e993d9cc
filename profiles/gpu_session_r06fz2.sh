#!/bin/bash
# the random-scene tests with all seven algorithms on the device + a further sweep; leaf sizes of the BVH on the mesh scene
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fuzz_scenes.py -q -m gpu 2>&1 | tail -3 > gpurun_out/r06fz2_pytest.log
timeout 400 python profiles/fuzz_sweep.py --first 400 --seconds 150 2>&1 | grep -v "switching from" > gpurun_out/r06fz2_sweep.log
VARIANTS="leaf1 leaf3 leaf4" REPS=2 bash profiles/m1_ab.sh > gpurun_out/r06fz2_leaf.txt 2>&1
tail -2 gpurun_out/r06fz2_pytest.log; tail -2 gpurun_out/r06fz2_sweep.log; cat gpurun_out/r06fz2_leaf.txt
