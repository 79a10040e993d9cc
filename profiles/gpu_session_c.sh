#!/bin/bash
# Session: new detmath on the device (numeric spec, parity), A/B against the previous commit's library.
set -u
TAG=${1:-r02d}
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/${TAG}
C=smallvcm_amd/csrc
B="python bench.py --no-cpu-baseline --no-configs --no-traffic --steps 12 --warmup 3"
run() { local name=$1; shift; echo "== $name" >> ${O}_ab.log; env "$@" 2>&1 | grep '^{' >> ${O}_ab.log; }
timeout 900 python -m pytest tests -q -m gpu --maxfail=10 > ${O}_pytest.log 2>&1
echo "pytest rc=$?" > ${O}_box.log; tail -4 ${O}_pytest.log
for rep in 1 2; do
run prev        SMALLVCM_AMD_LIB=$PWD/$C/libsmallvcm_amd_prev.so $B
run base        $B
done
run prev-512    SMALLVCM_AMD_LIB=$PWD/$C/libsmallvcm_amd_prev.so $B --res 512
run base-512    $B --res 512
run base-s3     $B --res 1024 --scene 3
run prev-s3     SMALLVCM_AMD_LIB=$PWD/$C/libsmallvcm_amd_prev.so $B --res 1024 --scene 3
run base-x2     $B --inflight 2
run base-hwq8-512x3 GPU_MAX_HW_QUEUES=8 $B --res 512 --inflight 3
run base-512x2  $B --res 512 --inflight 2
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
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats -- $B > ${O}_prof_bench.log 2>&1
