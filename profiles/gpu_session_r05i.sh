#!/bin/bash
# round 3, ninth call: the rectangle filter (SceneRects kernels) against the one-plane pair filter (SMALLVCM_AMD_NO_RECTS=1)
set -u
TAG=${1:-r05i}
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/${TAG}
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_kat.py tests/test_gpu_dropin_sharded.py -q -m gpu -x \
   -k "hip_equals_oracle or baseline_configs or other_box or long_paths or unmodified_reference or kat or late_iterations or path_tracer or determinism or full_size or reference_driver_over_dropin" \
   > ${O}_pytest.log 2>&1 < /dev/null; grep -E "passed|failed" ${O}_pytest.log | tail -2
B="python bench.py --no-cpu-baseline --no-configs --no-traffic --steps 12 --warmup 3"
run() { local name=$1; shift; echo "== $name" >> ${O}_ab.log; timeout 300 env "$@" 2>&1 < /dev/null | grep '^{' >> ${O}_ab.log; }
for rep in 1 2; do
  run quads-$rep SMALLVCM_AMD_NO_RECTS=1 $B
  run rects-$rep $B
done
run quads-bpm SMALLVCM_AMD_NO_RECTS=1 $B --algo bpm
run rects-bpm $B --algo bpm
run quads-bpt SMALLVCM_AMD_NO_RECTS=1 $B --algo bpt
run rects-bpt $B --algo bpt
run quads-512 SMALLVCM_AMD_NO_RECTS=1 $B --res 512
run rects-512 $B --res 512
run quads-s3 SMALLVCM_AMD_NO_RECTS=1 $B --scene 3 --res 1024
run rects-s3 $B --scene 3 --res 1024
python - ${TAG} <<'PY' | tee gpurun_out/${TAG}_ab_summary.txt
import json,sys
name=None
for l in open("gpurun_out/%s_ab.log" % sys.argv[1]):
    if l.startswith("=="): name=l[3:].strip(); continue
    d=json.loads(l); r=d["roofline"]["per_kernel"]
    print("%-20s %8.1f Mpaths/s  %7.3f ms/step  light %.3f cam %.3f conn %.3f merge %.3f" % (name, d["value"], d["ms_per_step"],
          r["k_light_trace"]["ms"], r["k_camera_trace"]["ms"], r["k_connect_di+vc"]["ms"], r["k_merge"]["ms"]))
PY
