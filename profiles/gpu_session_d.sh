#!/bin/bash
# Session: entry stamps, C++ farm, --report, drop-in throughput
set -u
TAG=${1:-r02f}
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/${TAG}
C=smallvcm_amd/csrc
B="python bench.py --no-cpu-baseline --no-configs --no-traffic --steps 12 --warmup 3"
run() { local name=$1; shift; echo "== $name" >> ${O}_ab.log; env "$@" 2>&1 | grep '^{' >> ${O}_ab.log; }
timeout 1500 python -m pytest tests/test_gpu_dropin_sharded.py -q -m gpu --maxfail=10 > ${O}_pytest_dropin.log 2>&1
echo "dropin rc=$?" > ${O}_box.log; tail -15 ${O}_pytest_dropin.log
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu --maxfail=10 -k "hip_equals_oracle or statistics or late or soak or C1 or C2" > ${O}_pytest_parity.log 2>&1
echo "parity rc=$?" >> ${O}_box.log; tail -3 ${O}_pytest_parity.log
for rep in 1 2; do
run prev        SMALLVCM_AMD_LIB=$PWD/$C/libsmallvcm_amd_prev.so $B
run base        $B
done
run nostamps    SMALLVCM_AMD_NO_STAMPS=1 $B
run prev-512    SMALLVCM_AMD_LIB=$PWD/$C/libsmallvcm_amd_prev.so $B --res 512
run base-512    $B --res 512
run nostamps-512 SMALLVCM_AMD_NO_STAMPS=1 $B --res 512
run base-512x2  $B --res 512 --inflight 2
run base-x2     $B --inflight 2
python - ${TAG} <<'PY' > ${O}_ab_summary.txt
import json,sys
name=None
for l in open("gpurun_out/%s_ab.log" % sys.argv[1]):
    if l.startswith("=="): name=l[3:].strip(); continue
    d=json.loads(l); r=d["roofline"]["per_kernel"]
    print("%-20s %8.1f Mpaths/s  %7.3f ms/step  light %.3f cam %.3f conn %.3f merge %.3f  iter %.3f sort %.3f grid %.3f" % (name, d["value"], d["ms_per_step"],
          r["k_light_trace"]["ms"], r["k_camera_trace"]["ms"], r["k_connect_di+vc"]["ms"], r["k_merge"]["ms"], d["roofline"]["iteration_ms"], d["roofline"]["query_sort_ms"], d["roofline"]["grid_build_ms_side_stream"]))
PY
cat ${O}_ab_summary.txt
# the reference's driver over the drop-in: 256 renderers (one per host core), 2048 iterations of 512^2
( time smallvcm_amd/dropin/smallvcm -s 1 -a vcm -i 2048 -o gpurun_out/${TAG}_dropin.hdr ) > ${O}_dropin_time.log 2>&1
tail -6 ${O}_dropin_time.log
smallvcm_amd/host/vcm_render -s 1 -a vcm -i 64 --res 2048 2048 --gpus 1 --shards 1 --inflight 2 --warmup 2 --json > ${O}_farm.log 2>&1
smallvcm_amd/host/vcm_render -s 1 -a vcm -i 64 --res 2048 2048 --renderers 1 --warmup 2 --json >> ${O}_farm.log 2>&1
cat ${O}_farm.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats -- $B > ${O}_prof_bench.log 2>&1
