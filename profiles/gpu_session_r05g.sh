#!/bin/bash
# round 3, seventh call: 64-byte aligned store / vertex records and one record per path slot, against the previous build
set -u
TAG=${1:-r05g}
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/${TAG}
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_scene2.py tests/test_gpu_dropin_sharded.py -q -m gpu -x \
   -k "hip_equals_oracle or C1-s1 or C2 or scene_file or mesh or merge_kernels or long_splat or sharded_contexts or render_farm or late_iterations or other_box or long_paths or path_tracer or cpp_farm or unmodified_reference" \
   > ${O}_pytest.log 2>&1 < /dev/null; grep -E "passed|failed" ${O}_pytest.log | tail -2
B="python bench.py --no-cpu-baseline --no-configs --no-traffic --steps 12 --warmup 3"
P=SMALLVCM_AMD_LIB=$PWD/smallvcm_amd/csrc/libsmallvcm_amd_prev.so
run() { local name=$1; shift; echo "== $name" >> ${O}_ab.log; timeout 300 env "$@" 2>&1 < /dev/null | grep '^{' >> ${O}_ab.log; }
run prev $P $B
run new $B
run prev2 $P $B
run new2 $B
run prev-bpm $P $B --algo bpm
run new-bpm $B --algo bpm
run prev-512 $P $B --res 512
run new-512 $B --res 512
run prev-s3 $P $B --scene 3 --res 1024
run new-s3 $B --scene 3 --res 1024
run prev-m1 $P $B --scene-file tests/scenes/bumpy_room.vcmscene --res 1024
run new-m1 $B --scene-file tests/scenes/bumpy_room.vcmscene --res 1024
python - ${TAG} <<'PY' | tee gpurun_out/${TAG}_ab_summary.txt
import json,sys
name=None
for l in open("gpurun_out/%s_ab.log" % sys.argv[1]):
    if l.startswith("=="): name=l[3:].strip(); continue
    d=json.loads(l); r=d["roofline"]["per_kernel"]
    print("%-20s %8.1f Mpaths/s  %7.3f ms/step  light %.3f cam %.3f conn %.3f merge %.3f" % (name, d["value"], d["ms_per_step"],
          r["k_light_trace"]["ms"], r["k_camera_trace"]["ms"], r["k_connect_di+vc"]["ms"], r["k_merge"]["ms"]))
PY
