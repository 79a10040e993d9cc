#!/bin/bash
# Average duration per launch of the big kernels under rocprofv3 --kernel-trace (kernels serialised per stream as the trace
# sees them), for the default build, variant builds (VARIANTS="a b": profiles/ab_<name>) and environment switches (ENVS="A=1 B=2"):
#   RES=2048 VARIANTS="abl1" ENVS="SMALLVCM_AMD_MERGE=walk" bash profiles/kstats.sh
cd "$(dirname "$0")/.." || exit 1
ROOT=$PWD
SCENE=${SCENE:-1}; ALGO=${ALGO:-vcm}; RES=${RES:-2048}; ITER=${ITER:-20}; WARM=${WARM:-5}
export TMPDIR=/tmp
one() {   # one <label> <exe> [env...]
  local label=$1 exe=$2; shift 2
  rm -rf /tmp/ks_run
  ( cd /tmp && env "$@" rocprofv3 --kernel-trace --output-format csv -d /tmp/ks_run -- $exe -s $SCENE -a $ALGO -i $ITER --warmup $WARM --res $RES $RES --json > /tmp/ks_run.json 2>/dev/null )
  python3 - "$label" $WARM <<'PY'
import csv, glob, collections, sys, json
label, warm = sys.argv[1], int(sys.argv[2])
f = glob.glob('/tmp/ks_run/**/*kernel_trace.csv', recursive=True)
d = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    n = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('vcm::', '').split('<')[0]
    d[n].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
def avg(n):
    v = d.get(n, [])
    v = v[len(v) // 4:]   # drop the warm-up quarter
    return sum(v) / len(v) if v else 0.0
names = ['k_merge_pairs', 'k_merge_walk', 'k_camera_trace', 'k_light_trace', 'k_connect_di', 'k_connect_vc', 'k_connect_camera', 'k_resolve', 'k_cell_rank_gather', 'k_cell_keys']
try: mp = json.loads(open('/tmp/ks_run.json').read().strip().splitlines()[-1])['Mpaths_s']
except Exception: mp = 0
print('%-36s %7.1f Mpaths/s | ' % (label, mp) + '  '.join('%s %.0f' % (n.replace('k_', ''), avg(n)) for n in names if avg(n) > 0))
PY
}
for rep in $(seq 1 ${REPS:-1}); do
  one default $ROOT/smallvcm_amd/host/vcm_render X=1
  for v in ${VARIANTS:-}; do one $v $ROOT/profiles/ab_$v/host/vcm_render X=1; done
  for e in ${ENVS:-}; do one "$e" $ROOT/smallvcm_amd/host/vcm_render ${e//,/ }; done
done
