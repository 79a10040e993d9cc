#!/bin/bash
# round 3, third call: two-launch scan without a serial stage (the single-pass scan was latency-bound), probe skip removed,
# scene files; A/B of the splat stream and of the persistent-wave count on small frames
set -u
TAG=${1:-r05c}
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/${TAG}
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_scene2.py tests/test_gpu_dropin_sharded.py -q -m gpu -x \
   -k "(hip_equals_oracle and wavefront) or C1-s1 or scene_file or bench_line or merge_kernels or long_splat or local_bbox or sharded_contexts" \
   > ${O}_pytest.log 2>&1 < /dev/null; grep -E "passed|failed" ${O}_pytest.log | tail -2
B="python bench.py --no-cpu-baseline --no-configs --no-traffic --steps 12 --warmup 3"
run() { local name=$1; shift; echo "== $name" >> ${O}_ab.log; timeout 300 env "$@" 2>&1 < /dev/null | grep '^{' >> ${O}_ab.log; }
run base $B
run splat1 SMALLVCM_AMD_SPLAT_STREAM=1 $B
run base2 $B
for R in 512 1024; do
  run base-$R $B --res $R
  run splat0-$R SMALLVCM_AMD_SPLAT_STREAM=0 $B --res $R
  for W in 1024 1536 2048 3072; do run waves$W-$R SMALLVCM_AMD_TRACE_WAVES=$W $B --res $R; done
done
run base-s3 $B --scene 3 --res 1024
run splat0-s3 SMALLVCM_AMD_SPLAT_STREAM=0 $B --scene 3 --res 1024
run waves2048-s3 SMALLVCM_AMD_TRACE_WAVES=2048 $B --scene 3 --res 1024
run base-bpm $B --algo bpm
python - ${TAG} <<'PY' | tee gpurun_out/${TAG}_ab_summary.txt
import json,sys
name=None
for l in open("gpurun_out/%s_ab.log" % sys.argv[1]):
    if l.startswith("=="): name=l[3:].strip(); continue
    d=json.loads(l); r=d["roofline"]["per_kernel"]
    print("%-20s %8.1f Mpaths/s  %7.3f ms/step  light %.3f cam %.3f conn %.3f merge %.3f" % (name, d["value"], d["ms_per_step"],
          r["k_light_trace"]["ms"], r["k_camera_trace"]["ms"], r["k_connect_di+vc"]["ms"], r["k_merge"]["ms"]))
PY
