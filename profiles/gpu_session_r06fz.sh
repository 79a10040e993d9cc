#!/bin/bash
# round 4, late: the random-scene tests on the device, a longer sweep of further seeds, the read-out-inclusive rate
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fuzz_scenes.py -q -m gpu 2>&1 | tail -5 > gpurun_out/r06fz_pytest.log
timeout 600 python profiles/fuzz_sweep.py --first 24 --seconds 300 --readout > gpurun_out/r06fz_sweep.log 2>&1
tail -3 gpurun_out/r06fz_pytest.log; tail -5 gpurun_out/r06fz_sweep.log
