#!/bin/bash
# round 3, sixth call: launch-shape A/B after the stream changes (K3c aside at 2048^2, task blocks), whole GPU suite
set -u
TAG=${1:-r05f}
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/${TAG}
timeout 1500 python -m pytest tests -q -m gpu -x > ${O}_pytest.log 2>&1 < /dev/null; grep -E "passed|failed" ${O}_pytest.log | tail -2
B="python bench.py --no-cpu-baseline --no-configs --no-traffic --steps 12 --warmup 3"
run() { local name=$1; shift; echo "== $name" >> ${O}_ab.log; timeout 300 env "$@" 2>&1 < /dev/null | grep '^{' >> ${O}_ab.log; }
run base $B
run vc1 SMALLVCM_AMD_VC_STREAM=1 $B
run tb2048 SMALLVCM_AMD_TASK_BLOCKS=2048 $B
run tb4608 SMALLVCM_AMD_TASK_BLOCKS=4608 $B
run base2 $B
run vc1-b SMALLVCM_AMD_VC_STREAM=1 $B
run noside SMALLVCM_AMD_NO_SIDE=1 $B
run base-bpm $B --algo bpm
run base-bpt $B --algo bpt
run base-512 $B --res 512
run base-512x4 $B --res 512 --inflight 4
run m1 $B --scene-file tests/scenes/bumpy_room.vcmscene --res 1024
python - ${TAG} <<'PY' | tee gpurun_out/${TAG}_ab_summary.txt
import json,sys
name=None
for l in open("gpurun_out/%s_ab.log" % sys.argv[1]):
    if l.startswith("=="): name=l[3:].strip(); continue
    d=json.loads(l); r=d["roofline"]["per_kernel"]
    print("%-20s %8.1f Mpaths/s  %7.3f ms/step  light %.3f cam %.3f conn %.3f merge %.3f" % (name, d["value"], d["ms_per_step"],
          r["k_light_trace"]["ms"], r["k_camera_trace"]["ms"], r["k_connect_di+vc"]["ms"], r["k_merge"]["ms"]))
PY
