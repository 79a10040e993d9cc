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
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu --maxfail=6 -k "(hip_equals_oracle and wavefront) or C1" > ${O}_pytest.log 2>&1
echo "pytest rc=$?" > ${O}_box.log; tail -3 ${O}_pytest.log
for rep in 1 2; do
run noboth-lane    SMALLVCM_AMD_LIB=$PWD/$C/libsmallvcm_amd_noboth.so SMALLVCM_AMD_MERGE=lane $B
run noocc-lane     SMALLVCM_AMD_LIB=$PWD/$C/libsmallvcm_amd_noocc.so SMALLVCM_AMD_MERGE=lane $B
run notab-lane     SMALLVCM_AMD_LIB=$PWD/$C/libsmallvcm_amd_notab.so SMALLVCM_AMD_MERGE=lane $B
run base-lane      SMALLVCM_AMD_MERGE=lane $B
done
run base-staged    SMALLVCM_AMD_MERGE=staged $B
run q32-lane       SMALLVCM_AMD_LIB=$PWD/$C/libsmallvcm_amd_q32.so SMALLVCM_AMD_MERGE=lane $B
run q24-lane       SMALLVCM_AMD_LIB=$PWD/$C/libsmallvcm_amd_q24.so SMALLVCM_AMD_MERGE=lane $B
run stage256       SMALLVCM_AMD_LIB=$PWD/$C/libsmallvcm_amd_stage256.so SMALLVCM_AMD_MERGE=staged $B
run base-lane3     SMALLVCM_AMD_MERGE=lane $B
run nostamps-2048  SMALLVCM_AMD_MERGE=lane SMALLVCM_AMD_NO_STAMPS=1 $B
run base-512       SMALLVCM_AMD_MERGE=lane $B --res 512
run nostamps-512   SMALLVCM_AMD_MERGE=lane SMALLVCM_AMD_NO_STAMPS=1 $B --res 512
run noboth-512     SMALLVCM_AMD_LIB=$PWD/$C/libsmallvcm_amd_noboth.so SMALLVCM_AMD_MERGE=lane $B --res 512
run inflight2-512  SMALLVCM_AMD_MERGE=lane $B --res 512 --inflight 2
run inflight3-512  SMALLVCM_AMD_MERGE=lane $B --res 512 --inflight 3
run inflight4a2-512 SMALLVCM_AMD_MERGE=lane SMALLVCM_AMD_ARENAS=2 $B --res 512 --inflight 4
run inflight2ns-512 SMALLVCM_AMD_MERGE=lane SMALLVCM_AMD_NO_STAMPS=1 $B --res 512 --inflight 2
run inflight3-2048 SMALLVCM_AMD_MERGE=lane $B --inflight 3
python - ${TAG} <<'PY' > ${O}_ab_summary.txt
import json,sys
name=None
for l in open("gpurun_out/%s_ab.log" % sys.argv[1]):
    if l.startswith("=="): name=l[3:].strip(); continue
    d=json.loads(l); r=d["roofline"]["per_kernel"]
    print("%-20s %8.1f Mpaths/s  %7.3f ms/step  light %.3f cam %.3f conn %.3f merge %.3f  iter %.3f" % (name, d["value"], d["ms_per_step"],
          r["k_light_trace"]["ms"], r["k_camera_trace"]["ms"], r["k_connect_di+vc"]["ms"], r["k_merge"]["ms"], d["roofline"]["iteration_ms"]))
PY
cat ${O}_ab_summary.txt
SMALLVCM_AMD_MERGE=lane timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats -- $B > ${O}_prof_bench.log 2>&1
