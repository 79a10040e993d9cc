#!/bin/bash
# round 3, tenth call: four-wide BVH nodes against the two-wide ones (bvh2 = the previous commit's library)
set -u
TAG=${1:-r05j}
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/${TAG}
timeout 1200 python -m pytest tests/test_scene2.py tests/test_gpu_parity.py -q -m gpu -x -k "mesh or forced_bvh or scene_file or tilted or (hip_equals_oracle and wavefront and 64)" > ${O}_pytest.log 2>&1 < /dev/null; grep -E "passed|failed" ${O}_pytest.log | tail -2
B="python bench.py --no-cpu-baseline --no-configs --no-traffic --steps 12 --warmup 3 --scene-file tests/scenes/bumpy_room.vcmscene"
L=$PWD/smallvcm_amd/csrc
run() { local name=$1; shift; echo "== $name" >> ${O}_ab.log; timeout 300 env "$@" 2>&1 < /dev/null | grep '^{' >> ${O}_ab.log; }
for rep in 1 2; do
  run bvh2-$rep SMALLVCM_AMD_LIB=$L/libsmallvcm_amd_bvh2.so $B --res 1024
  run bvh4-$rep $B --res 1024
  run bvh4k3w4-$rep SMALLVCM_AMD_LIB=$L/libsmallvcm_amd_k3w4.so $B --res 1024
done
run bvh2-512 SMALLVCM_AMD_LIB=$L/libsmallvcm_amd_bvh2.so $B --res 512
run bvh4-512 $B --res 512
run bvh2-bpt SMALLVCM_AMD_LIB=$L/libsmallvcm_amd_bvh2.so $B --res 1024 --algo bpt
run bvh4-bpt $B --res 1024 --algo bpt
python - ${TAG} <<'PY' | tee gpurun_out/${TAG}_ab_summary.txt
import json,sys
name=None
for l in open("gpurun_out/%s_ab.log" % sys.argv[1]):
    if l.startswith("=="): name=l[3:].strip(); continue
    d=json.loads(l); r=d["roofline"]["per_kernel"]
    print("%-20s %8.1f Mpaths/s  %7.3f ms/step  light %.3f cam %.3f conn %.3f merge %.3f" % (name, d["value"], d["ms_per_step"],
          r["k_light_trace"]["ms"], r["k_camera_trace"]["ms"], r["k_connect_di+vc"]["ms"], r["k_merge"]["ms"]))
PY
