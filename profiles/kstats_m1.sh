#!/bin/bash
# per-kernel totals of the scene-file configuration M1 (10 380 primitives, BVH, VCM 1024^2): rocprofv3 --kernel-trace --stats
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
rm -rf /tmp/ks_m1; (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_m1 -- $OLDPWD/smallvcm_amd/host/vcm_render --scene-file $OLDPWD/tests/scenes/bumpy_room.vcmscene -a vcm -i 10 --warmup 2 --res 1024 1024 --json > /tmp/ks_m1.json 2>/dev/null)
python3 - <<'PY'
import csv,glob,json
f=glob.glob('/tmp/ks_m1/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r['Name'].split('(')[0].replace('void ','').replace('vcm::','')
    t=float(r['TotalDurationNs'])/1e6
    if t>0.3: print('%-44s %8.2f ms total  %6.1f us avg  %s calls'%(n[:44],t,float(r['AverageNs'])/1e3,r['Calls']))
print(json.loads(open('/tmp/ks_m1.json').read().strip().splitlines()[-1]))
PY
