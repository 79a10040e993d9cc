#!/bin/bash
# round 3, eighth call: which of the two record splits pays?  new = both split; store80 / vq80 = one of them back to the
# 80-byte record; both80 = round 2's layout (= prev without the path-slot records that r05g showed to be slower)
set -u
TAG=${1:-r05h}
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/${TAG}
B="python bench.py --no-cpu-baseline --no-configs --no-traffic --steps 12 --warmup 3"
L=$PWD/smallvcm_amd/csrc
run() { local name=$1; shift; echo "== $name" >> ${O}_ab.log; timeout 300 env "$@" 2>&1 < /dev/null | grep '^{' >> ${O}_ab.log; }
for rep in 1 2; do
  run both80-$rep SMALLVCM_AMD_LIB=$L/libsmallvcm_amd_both80.so $B
  run new-$rep $B
  run store80-$rep SMALLVCM_AMD_LIB=$L/libsmallvcm_amd_store80.so $B
  run vq80-$rep SMALLVCM_AMD_LIB=$L/libsmallvcm_amd_vq80.so $B
done
run both80-bpm SMALLVCM_AMD_LIB=$L/libsmallvcm_amd_both80.so $B --algo bpm
run new-bpm $B --algo bpm
run store80-bpm SMALLVCM_AMD_LIB=$L/libsmallvcm_amd_store80.so $B --algo bpm
run vq80-bpm SMALLVCM_AMD_LIB=$L/libsmallvcm_amd_vq80.so $B --algo bpm
run both80-s3 SMALLVCM_AMD_LIB=$L/libsmallvcm_amd_both80.so $B --scene 3 --res 1024
run new-s3 $B --scene 3 --res 1024
run store80-s3 SMALLVCM_AMD_LIB=$L/libsmallvcm_amd_store80.so $B --scene 3 --res 1024
run vq80-s3 SMALLVCM_AMD_LIB=$L/libsmallvcm_amd_vq80.so $B --scene 3 --res 1024
python - ${TAG} <<'PY' | tee gpurun_out/${TAG}_ab_summary.txt
import json,sys
name=None
for l in open("gpurun_out/%s_ab.log" % sys.argv[1]):
    if l.startswith("=="): name=l[3:].strip(); continue
    d=json.loads(l); r=d["roofline"]["per_kernel"]
    print("%-20s %8.1f Mpaths/s  %7.3f ms/step  light %.3f cam %.3f conn %.3f merge %.3f" % (name, d["value"], d["ms_per_step"],
          r["k_light_trace"]["ms"], r["k_camera_trace"]["ms"], r["k_connect_di+vc"]["ms"], r["k_merge"]["ms"]))
PY
