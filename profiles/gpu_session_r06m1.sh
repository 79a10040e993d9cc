#!/bin/bash
# BVH changes: the mesh-scene and random-scene tests on the device, then M1 without a profiler attached
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_scene2.py tests/test_fuzz_scenes.py tests/test_kat.py -q -m gpu 2>&1 | tail -4 > gpurun_out/r06m1_pytest.log
for i in 1 2 3; do smallvcm_amd/host/vcm_render --scene-file tests/scenes/bumpy_room.vcmscene -a vcm -i 40 --warmup 5 --res 1024 1024 --json 2>/dev/null | tail -1; done > gpurun_out/r06m1_rate.log
cat gpurun_out/r06m1_pytest.log; python3 - <<'PY'
import json
for l in open('gpurun_out/r06m1_rate.log'):
    d = json.loads(l); print('M1', d['Mpaths_s'], 'Mpaths/s', d['last_iteration_ms'], 'ms')
PY
