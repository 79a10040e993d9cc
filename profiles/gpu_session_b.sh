#!/bin/bash
# Session r02b: A/B of library variants (SMALLVCM_AMD_LIB), merge kernels, renderers in flight.
set -u
TAG=${1:-r02b}
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/${TAG}
C=smallvcm_amd/csrc
B="python bench.py --no-cpu-baseline --no-configs --no-traffic --steps 12 --warmup 3"
run() { # name, env..., -- args
  local name=$1; shift
  echo "== $name" >> ${O}_ab.log
  env "$@" 2>&1 | grep '^{' >> ${O}_ab.log
}
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_kat.py tests/test_gpu_dropin_sharded.py -q -m gpu --maxfail=6 \
   -k "(hip_equals_oracle and wavefront) or kat or interleaving or concurrent or render_farm or long_paths or C1" > ${O}_pytest.log 2>&1
echo "pytest rc=$?" > ${O}_box.log; tail -3 ${O}_pytest.log
run prev-lane      SMALLVCM_AMD_LIB=$PWD/$C/libsmallvcm_amd_prev.so SMALLVCM_AMD_MERGE=lane $B
run base-lane      SMALLVCM_AMD_MERGE=lane $B
run base-staged    SMALLVCM_AMD_MERGE=staged $B
run q32-lane       SMALLVCM_AMD_LIB=$PWD/$C/libsmallvcm_amd_q32.so SMALLVCM_AMD_MERGE=lane $B
run q24-lane       SMALLVCM_AMD_LIB=$PWD/$C/libsmallvcm_amd_q24.so SMALLVCM_AMD_MERGE=lane $B
run stage256       SMALLVCM_AMD_LIB=$PWD/$C/libsmallvcm_amd_stage256.so SMALLVCM_AMD_MERGE=staged $B
run prev-lane2     SMALLVCM_AMD_LIB=$PWD/$C/libsmallvcm_amd_prev.so SMALLVCM_AMD_MERGE=lane $B
run base-lane2     SMALLVCM_AMD_MERGE=lane $B
run lane-chunk4    SMALLVCM_AMD_MERGE=lane SMALLVCM_AMD_MERGE_CHUNK=4 $B
run lane-chunk64   SMALLVCM_AMD_MERGE=lane SMALLVCM_AMD_MERGE_CHUNK=64 $B
run inflight2-2048 SMALLVCM_AMD_MERGE=lane $B --inflight 2
for w in 4096 2048 1024; do run w$w-512 SMALLVCM_AMD_MERGE=lane SMALLVCM_AMD_TRACE_WAVES=$w $B --res 512; done
for k in 2 4 8; do run inflight$k-512 SMALLVCM_AMD_MERGE=lane $B --res 512 --inflight $k; done
run inflight2-s3-1024 SMALLVCM_AMD_MERGE=lane $B --res 1024 --scene 3 --inflight 2
run s3-1024 SMALLVCM_AMD_MERGE=lane $B --res 1024 --scene 3
python - <<'PY' > ${O}_ab_summary.txt
import json,sys
name=None
for l in open("gpurun_out/%s_ab.log" % "${TAG}"):
    if l.startswith("=="): name=l[3:].strip(); continue
    d=json.loads(l); r=d["roofline"]["per_kernel"]
    print("%-20s %8.1f Mpaths/s  %7.3f ms/step  light %.3f cam %.3f conn %.3f merge %.3f  iter %.3f" % (name, d["value"], d["ms_per_step"],
          r["k_light_trace"]["ms"], r["k_camera_trace"]["ms"], r["k_connect_di+vc"]["ms"], r["k_merge"]["ms"], d["roofline"]["iteration_ms"]))
PY
cat ${O}_ab_summary.txt
SMALLVCM_AMD_MERGE=lane timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats -- $B > ${O}_prof_bench.log 2>&1
