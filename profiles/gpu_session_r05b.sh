#!/bin/bash
# round 3, second call: single-pass scan, splat stream, edge/corner probe skip in k_merge_walk, folded one-lane kernels:
# the whole GPU suite, then A/B lines
set -u
TAG=${1:-r05b}
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/${TAG}
timeout 1200 python -m pytest tests -q -m gpu -x > ${O}_pytest.log 2>&1 < /dev/null; tail -3 ${O}_pytest.log
B="python bench.py --no-cpu-baseline --no-configs --no-traffic --steps 12 --warmup 3"
run() { local name=$1; shift; echo "== $name" >> ${O}_ab.log; timeout 300 env "$@" 2>&1 < /dev/null | grep '^{' >> ${O}_ab.log; }
run base $B
run noskip SMALLVCM_AMD_LIB=$PWD/smallvcm_amd/csrc/libsmallvcm_amd_noskip.so $B
run base2 $B
run splat1 SMALLVCM_AMD_SPLAT_STREAM=1 $B
for R in 512 1024; do
  run base-$R $B --res $R
  run splat0-$R SMALLVCM_AMD_SPLAT_STREAM=0 $B --res $R
done
run waves2048-512 SMALLVCM_AMD_TRACE_WAVES=2048 $B --res 512
run waves3072-512 SMALLVCM_AMD_TRACE_WAVES=3072 $B --res 512
run base-s3 $B --scene 3 --res 1024
run splat0-s3 SMALLVCM_AMD_SPLAT_STREAM=0 $B --scene 3 --res 1024
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
