#!/bin/bash
# A/B only (no parity suite): default build, every variant library, and the env switches given as extra arguments "name:VAR=val"
set -u
TAG=${1:-r02k}; shift
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/${TAG}
B="python bench.py --no-cpu-baseline --no-configs --no-traffic --steps 12 --warmup 3"
run() { local name=$1; shift; echo "== $name" >> ${O}_ab.log; env "$@" 2>&1 | grep '^{' >> ${O}_ab.log; }
run base $B
for lib in smallvcm_amd/csrc/libsmallvcm_amd_*.so; do
  [ -f "$lib" ] || continue
  n=$(basename $lib .so); n=${n#libsmallvcm_amd_}
  run $n SMALLVCM_AMD_LIB=$lib $B
done
for sw in "$@"; do run "${sw%%:*}" "${sw#*:}" $B; done
run base2 $B
run base-512 $B --res 512
run base-s3 $B --scene 3 --res 1024
run base-bpm $B --algo bpm
for lib in smallvcm_amd/csrc/libsmallvcm_amd_*.so; do
  [ -f "$lib" ] || continue
  n=$(basename $lib .so); n=${n#libsmallvcm_amd_}
  run $n-512 SMALLVCM_AMD_LIB=$lib $B --res 512
done
python - ${TAG} <<'PY' | tee gpurun_out/${TAG}_ab_summary.txt
import json,sys
name=None
for l in open("gpurun_out/%s_ab.log" % sys.argv[1]):
    if l.startswith("=="): name=l[3:].strip(); continue
    d=json.loads(l); r=d["roofline"]["per_kernel"]
    print("%-20s %8.1f Mpaths/s  %7.3f ms/step  light %.3f cam %.3f conn %.3f merge %.3f" % (name, d["value"], d["ms_per_step"],
          r["k_light_trace"]["ms"], r["k_camera_trace"]["ms"], r["k_connect_di+vc"]["ms"], r["k_merge"]["ms"]))
PY
