#!/bin/bash
# v2 scenes: parity of the BVH / list paths, throughput of the mesh scenes with the default build and every variant
set -u
TAG=${1:-r02v}
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/${TAG}
timeout 900 python -m pytest tests/test_scene2.py tests/test_kat.py tests/test_gpu_parity.py -q -m gpu -x -k "scene or kat or hip_equals_oracle or other_box or long_paths or C1-s1" > ${O}_pytest_scene2.log 2>&1 < /dev/null; tail -3 ${O}_pytest_scene2.log
mesh() {
python - "$1" <<'PY'
import sys, time
sys.path.insert(0, "tests")
from mesh_scenes import bumpy_room, tilted_room
from smallvcm_amd.renderer import VertexCM
for name, sc, res in (("bumpy room grid 72", bumpy_room(grid=72, resx=1024, resy=1024), 1024), ("bumpy room grid 200", bumpy_room(grid=200, resx=1024, resy=1024), 1024),
                      ("tilted room (list)", tilted_room(resx=1024, resy=1024), 1024)):
    r = VertexCM(sc, 4, 0.003, 0.75, 1234); r.mMaxPathLength = 10
    for it in range(3): r.RunIteration(it)
    r.backend.synchronize(); t = time.perf_counter()
    for it in range(3, 11): r.RunIteration(it)
    r.backend.synchronize(); dt = time.perf_counter() - t
    st = r.stats()
    print("%-10s %-22s %6d primitives, %dx%d vcm: %7.1f Mpaths/s (%.2f ms / iteration; light %.2f camera %.2f connect %.2f merge %.2f)" % (sys.argv[1], name, sc.nPrims, res, res,
          2 * res * res * 8 / dt / 1e6, dt / 8 * 1e3, st["msLightKernel"], st["msCameraKernel"], st["msConnectKernels"], st["msMergeKernel"]))
    r.close()
PY
}
mesh default | tee ${O}_mesh.txt
for lib in smallvcm_amd/csrc/libsmallvcm_amd_*.so; do
  [ -f "$lib" ] || continue
  n=$(basename $lib .so); n=${n#libsmallvcm_amd_}
  SMALLVCM_AMD_LIB=$lib mesh $n | tee -a ${O}_mesh.txt
done
