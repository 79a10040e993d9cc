#!/bin/bash
# K4 per rank with 8 serialised shards at 2048^2 (scene 1 VCM): mean / min / max over ranks and iterations, index mode and space mode
#   bash profiles/tools/k4_shards.sh
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
run() {   # run <label> env...
  local label=$1; shift
  rm -rf /tmp/k4s8
  (cd /tmp && env "$@" SMALLVCM_AMD_FARM_SERIALIZE=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/k4s8 -- $OLDPWD/smallvcm_amd/host/vcm_render -s 1 -a vcm -i 6 --warmup 2 --res 2048 2048 --gpus 8 --shards 8 --inflight 1 --collectives threads --json > /dev/null 2>&1)
  python3 - "$label" <<'PY'
import csv,glob,sys,collections
f=glob.glob('/tmp/k4s8/**/*kernel_trace.csv',recursive=True)
per=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    n=r['Kernel_Name'].split('(')[0].replace('void ','').replace('vcm::','').split('<')[0]
    per[n].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
def st(n):
    d=per.get(n,[]); d=d[len(d)//4:]
    return 'mean %.0f min %.0f max %.0f'%(sum(d)/len(d),min(d),max(d)) if d else '-'
print('%-28s k_merge_pairs %s | k_space_scatter %s | k_camera_trace %s'%(sys.argv[1], st('k_merge_pairs'), st('k_space_scatter'), st('k_camera_trace')))
PY
}
run index SMALLVCM_AMD_FARM_MERGE=index
run space SMALLVCM_AMD_FARM_MERGE=space
