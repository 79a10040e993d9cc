#!/bin/bash
set -u
TAG=${1:-r02z}
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/${TAG}
timeout 600 python -m pytest tests/test_scene2.py tests/test_gpu_parity.py -q -m gpu -x -k "mesh or tilted or hip_equals_oracle or other_box or C1-s1" > ${O}_pytest.log 2>&1 < /dev/null; tail -2 ${O}_pytest.log
B="python bench.py --no-cpu-baseline --no-configs --no-traffic --steps 12 --warmup 3"
for n in base base2; do timeout 200 $B 2>&1 < /dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$n %.1f Mpaths/s %.3f ms' % (d['value'], d['ms_per_step']))"; done | tee ${O}_bench.txt
timeout 300 python - <<'PY' 2>&1 | tee ${O}_mesh.txt
import sys, time
sys.path.insert(0, "tests")
from mesh_scenes import bumpy_room, tilted_room
from smallvcm_amd.renderer import VertexCM
for name, sc, res in (("bumpy room grid 72", bumpy_room(grid=72, resx=1024, resy=1024), 1024), ("tilted room (list)", tilted_room(resx=1024, resy=1024), 1024)):
    r = VertexCM(sc, 4, 0.003, 0.75, 1234); r.mMaxPathLength = 10
    for it in range(3): r.RunIteration(it)
    r.backend.synchronize(); t = time.perf_counter()
    for it in range(3, 11): r.RunIteration(it)
    r.backend.synchronize(); dt = time.perf_counter() - t
    print("%-22s %6d primitives, %dx%d vcm: %7.1f Mpaths/s (%.2f ms / iteration)" % (name, sc.nPrims, res, res, 2 * res * res * 8 / dt / 1e6, dt / 8 * 1e3))
    r.close()
PY
