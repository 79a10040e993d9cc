#!/bin/bash
set -u
TAG=${1:-r02g}
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/${TAG}
B="python bench.py --no-cpu-baseline --no-configs --no-traffic --steps 12 --warmup 3"
run() { local name=$1; shift; echo "== $name" >> ${O}_ab.log; env "$@" 2>&1 | grep '^{' >> ${O}_ab.log; }
timeout 2400 python -m pytest tests -q -m gpu --maxfail=15 > ${O}_pytest.log 2>&1
echo "pytest rc=$?" > ${O}_box.log; tail -25 ${O}_pytest.log
run base $B
run base-512 $B --res 512
python - ${TAG} <<'PY'
import json,sys
name=None
for l in open("gpurun_out/%s_ab.log" % sys.argv[1]):
    if l.startswith("=="): name=l[3:].strip(); continue
    d=json.loads(l); r=d["roofline"]["per_kernel"]
    print("%-20s %8.1f Mpaths/s  %7.3f ms/step  light %.3f cam %.3f conn %.3f merge %.3f" % (name, d["value"], d["ms_per_step"],
          r["k_light_trace"]["ms"], r["k_camera_trace"]["ms"], r["k_connect_di+vc"]["ms"], r["k_merge"]["ms"]))
PY
# a large mesh scene through the BVH: throughput
python - <<'PY' > ${O}_mesh.log 2>&1
import sys, time
sys.path.insert(0, "tests")
from mesh_scenes import bumpy_room
from smallvcm_amd.renderer import VertexCM
for grid, res in ((72, 1024), (200, 1024)):
    sc = bumpy_room(grid=grid, resx=res, resy=res)
    r = VertexCM(sc, 4, 0.003, 0.75, 1234); r.mMaxPathLength = 10
    for it in range(3): r.RunIteration(it)
    r.backend.synchronize(); t = time.perf_counter()
    for it in range(3, 11): r.RunIteration(it)
    r.backend.synchronize(); dt = time.perf_counter() - t
    print("bumpy room, %d primitives, %dx%d vcm: %.1f Mpaths/s (%.2f ms / iteration)" % (sc.nPrims, res, res, 2 * res * res * 8 / dt / 1e6, dt / 8 * 1e3), r.stats())
    r.close()
PY
cat ${O}_mesh.log
