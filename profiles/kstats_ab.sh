#!/bin/bash
# per-kernel times of the default build and of the cross-revision baseline (profiles/make_ab_base.sh) for one configuration:
# rocprofv3 --kernel-trace --stats around vcm_render, total ms per kernel over ITER iterations, side by side.
#   SCENE=3 RES=1024 bash profiles/kstats_ab.sh
cd "$(dirname "$0")/.." || exit 1
SCENE=${SCENE:-1}; ALGO=${ALGO:-vcm}; RES=${RES:-2048}; ITER=${ITER:-10}; WARM=${WARM:-2}
export TMPDIR=/tmp
for w in default base; do
  exe=smallvcm_amd/host/vcm_render; [ $w = base ] && exe=profiles/ab_base/host/vcm_render
  rm -rf /tmp/ks_$w; (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$w -- $OLDPWD/$exe -s $SCENE -a $ALGO -i $ITER --warmup $WARM --res $RES $RES --json > /tmp/ks_$w.json 2>/dev/null)
done
python3 - <<'PY'
import csv,glob,collections
def load(w):
    f=glob.glob('/tmp/ks_%s/**/*kernel_stats.csv'%w,recursive=True)
    d=collections.OrderedDict()
    for r in csv.DictReader(open(f[0])):
        n=r['Name'].split('(')[0].replace('void ','').replace('vcm::','')
        d[n]=(float(r['TotalDurationNs'])/1e6,int(r['Calls']))
    return d
a,b=load('default'),load('base')
keys=list(a)+[k for k in b if k not in a]
print('%-44s %10s %10s'%('kernel (total ms over the run)','default','base'))
ta=tb=0
for k in keys:
    x=a.get(k,(0,0))[0]; y=b.get(k,(0,0))[0]; ta+=x; tb+=y
    if max(x,y)>0.05: print('%-44s %10.2f %10.2f'%(k[:44],x,y))
print('%-44s %10.2f %10.2f'%('sum',ta,tb))
import json
for w in ('default','base'):
    try: print(w, json.loads(open('/tmp/ks_%s.json'%w).read().strip().splitlines()[-1])['Mpaths_s'])
    except Exception as e: print(w,'?',e)
PY
