#!/bin/bash
set -u
TAG=${1:-r03e}; shift
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/${TAG}
timeout 500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "hip_equals_oracle or path_tracer or C1-s1 or long_paths" > ${O}_pytest.log 2>&1 < /dev/null; tail -2 ${O}_pytest.log
B="python bench.py --no-cpu-baseline --no-configs --no-traffic --steps 12 --warmup 3"
run() { local name=$1; shift; echo "== $name" >> ${O}_ab.log; timeout 200 env "$@" 2>&1 < /dev/null | grep '^{' >> ${O}_ab.log; }
run base $B
for sw in "$@"; do run "${sw%%:*}" "${sw#*:}" $B; done
run base2 $B
run base-512 $B --res 512
for sw in "$@"; do run "${sw%%:*}-512" "${sw#*:}" $B --res 512; done
run base-s3 $B --scene 3 --res 1024
python - ${TAG} <<'PY' | tee gpurun_out/${TAG}_ab_summary.txt
import json,sys
name=None
for l in open("gpurun_out/%s_ab.log" % sys.argv[1]):
    if l.startswith("=="): name=l[3:].strip(); continue
    d=json.loads(l); r=d["roofline"]["per_kernel"]
    print("%-20s %8.1f Mpaths/s  %7.3f ms/step  light %.3f cam %.3f conn %.3f merge %.3f" % (name, d["value"], d["ms_per_step"],
          r["k_light_trace"]["ms"], r["k_camera_trace"]["ms"], r["k_connect_di+vc"]["ms"], r["k_merge"]["ms"]))
PY
