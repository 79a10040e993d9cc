#!/bin/bash
# occupancy experiments: K1/K3/K4 capped to 5 waves per SIMD (spills to scratch), with the matching number of persistent waves
set -u
TAG=${1:-r02s}
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/${TAG}
B="python bench.py --no-cpu-baseline --no-configs --no-traffic --steps 12 --warmup 3"
run() { local name=$1; shift; echo "== $name" >> ${O}_ab.log; env "$@" 2>&1 | grep '^{' >> ${O}_ab.log; }
L=smallvcm_amd/csrc
run base $B
run k3w5-4096 SMALLVCM_AMD_LIB=$L/libsmallvcm_amd_k3w5.so $B
run k3w5-5120 SMALLVCM_AMD_LIB=$L/libsmallvcm_amd_k3w5.so SMALLVCM_AMD_TRACE_WAVES=5120 SMALLVCM_AMD_LIGHT_WAVES=4096 $B
run k13w5-5120 SMALLVCM_AMD_LIB=$L/libsmallvcm_amd_k13w5.so SMALLVCM_AMD_TRACE_WAVES=5120 $B
run k4w5 SMALLVCM_AMD_LIB=$L/libsmallvcm_amd_k4w5.so $B
run base2 $B
run k13w5-5120-512 SMALLVCM_AMD_LIB=$L/libsmallvcm_amd_k13w5.so SMALLVCM_AMD_TRACE_WAVES=5120 $B --res 512
run base-512 $B --res 512
python - ${TAG} <<'PY' | tee gpurun_out/${TAG}_ab_summary.txt
import json,sys
name=None
for l in open("gpurun_out/%s_ab.log" % sys.argv[1]):
    if l.startswith("=="): name=l[3:].strip(); continue
    d=json.loads(l); r=d["roofline"]["per_kernel"]
    print("%-20s %8.1f Mpaths/s  %7.3f ms/step  light %.3f cam %.3f conn %.3f merge %.3f" % (name, d["value"], d["ms_per_step"],
          r["k_light_trace"]["ms"], r["k_camera_trace"]["ms"], r["k_connect_di+vc"]["ms"], r["k_merge"]["ms"]))
PY
