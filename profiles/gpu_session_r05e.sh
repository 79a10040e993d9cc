#!/bin/bash
# round 3, fifth call: K4 batches dealt as per-XCD slabs with stealing (A/B), BVH leaves in leaf order
set -u
TAG=${1:-r05e}
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/${TAG}
timeout 900 env SMALLVCM_AMD_MERGE_DEAL=slab python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "(hip_equals_oracle and wavefront) or merge_kernels or C1-s1 or late_iterations" \
   > ${O}_pytest_slab.log 2>&1 < /dev/null; grep -E "passed|failed" ${O}_pytest_slab.log | tail -2
timeout 900 python -m pytest tests/test_scene2.py -q -m gpu -x > ${O}_pytest.log 2>&1 < /dev/null; grep -E "passed|failed" ${O}_pytest.log | tail -2
B="python bench.py --no-cpu-baseline --no-configs --no-traffic --steps 12 --warmup 3"
run() { local name=$1; shift; echo "== $name" >> ${O}_ab.log; timeout 300 env "$@" 2>&1 < /dev/null | grep '^{' >> ${O}_ab.log; }
run base $B
run slab SMALLVCM_AMD_MERGE_DEAL=slab $B
run slab2048 SMALLVCM_AMD_MERGE_DEAL=slab SMALLVCM_AMD_MERGE_SLAB_BLOCKS=2048 $B
run base2 $B
run slab-b SMALLVCM_AMD_MERGE_DEAL=slab $B
run base-bpm $B --algo bpm
run slab-bpm SMALLVCM_AMD_MERGE_DEAL=slab $B --algo bpm
run base-512 $B --res 512
run slab-512 SMALLVCM_AMD_MERGE_DEAL=slab $B --res 512
run base-1024 $B --res 1024
run slab-1024 SMALLVCM_AMD_MERGE_DEAL=slab $B --res 1024
run m1 $B --scene-file tests/scenes/bumpy_room.vcmscene --res 1024
python - ${TAG} <<'PY' | tee gpurun_out/${TAG}_ab_summary.txt
import json,sys
name=None
for l in open("gpurun_out/%s_ab.log" % sys.argv[1]):
    if l.startswith("=="): name=l[3:].strip(); continue
    d=json.loads(l); r=d["roofline"]["per_kernel"]
    print("%-20s %8.1f Mpaths/s  %7.3f ms/step  light %.3f cam %.3f conn %.3f merge %.3f" % (name, d["value"], d["ms_per_step"],
          r["k_light_trace"]["ms"], r["k_camera_trace"]["ms"], r["k_connect_di+vc"]["ms"], r["k_merge"]["ms"]))
PY
# HBM traffic of K4 under both dealings (one counter pass each)
for M in chunk slab; do
  for C in FETCH_SIZE; do
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/${TAG}_pmc_${M} -- env SMALLVCM_AMD_MERGE_DEAL=$M $B --child > /dev/null 2>&1
  done
done
python - ${TAG} <<'PY' | tee -a gpurun_out/${TAG}_ab_summary.txt
import csv,glob,sys
for m in ("chunk","slab"):
    f=glob.glob("gpurun_out/%s_pmc_%s/**/*counter_collection.csv" % (sys.argv[1], m), recursive=True)
    if not f: print(m, "no csv"); continue
    v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if "k_merge_walk" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE"]
    v=v[3:]
    print("k_merge_walk %-6s FETCH_SIZE x2 = %.2f GB per launch (%d launches)" % (m, 2*1024*sum(v)/len(v)/1e9, len(v)))
PY
