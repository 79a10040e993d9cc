#!/bin/bash
# round 3, fourth call: wide-node while-while BVH traversal, K3c on the splat stream on small frames, splat stream always on
set -u
TAG=${1:-r05d}
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/${TAG}
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_scene2.py tests/test_gpu_dropin_sharded.py -q -m gpu -x \
   -k "(hip_equals_oracle and wavefront) or C1-s1 or C2 or scene_file or mesh or forced_bvh or tilted or path_tracer or long_splat or sharded_contexts or render_farm or interleaving or concurrent" \
   > ${O}_pytest.log 2>&1 < /dev/null; grep -E "passed|failed" ${O}_pytest.log | tail -2
B="python bench.py --no-cpu-baseline --no-configs --no-traffic --steps 12 --warmup 3"
run() { local name=$1; shift; echo "== $name" >> ${O}_ab.log; timeout 300 env "$@" 2>&1 < /dev/null | grep '^{' >> ${O}_ab.log; }
run base $B
run splat0 SMALLVCM_AMD_SPLAT_STREAM=0 $B
run vc1 SMALLVCM_AMD_VC_STREAM=1 $B
run base2 $B
for R in 512 1024; do
  run base-$R $B --res $R
  run vc0-$R SMALLVCM_AMD_VC_STREAM=0 $B --res $R
done
run base-s3 $B --scene 3 --res 1024
run vc0-s3 SMALLVCM_AMD_VC_STREAM=0 $B --scene 3 --res 1024
run base-bpm $B --algo bpm
run m1 $B --scene-file tests/scenes/bumpy_room.vcmscene --res 1024
run m1-threaded SMALLVCM_AMD_LIB=$PWD/smallvcm_amd/csrc/libsmallvcm_amd_threaded.so $B --scene-file tests/scenes/bumpy_room.vcmscene --res 1024
run m1-512 $B --scene-file tests/scenes/bumpy_room.vcmscene --res 512
python - ${TAG} <<'PY' | tee gpurun_out/${TAG}_ab_summary.txt
import json,sys
name=None
for l in open("gpurun_out/%s_ab.log" % sys.argv[1]):
    if l.startswith("=="): name=l[3:].strip(); continue
    d=json.loads(l); r=d["roofline"]["per_kernel"]
    print("%-20s %8.1f Mpaths/s  %7.3f ms/step  light %.3f cam %.3f conn %.3f merge %.3f" % (name, d["value"], d["ms_per_step"],
          r["k_light_trace"]["ms"], r["k_camera_trace"]["ms"], r["k_connect_di+vc"]["ms"], r["k_merge"]["ms"]))
PY
