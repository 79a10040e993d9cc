#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per kernel of the default build and of the cross-revision baseline (serialised by the counter pass):
# average duration and raw counter sums per launch.   KERNELS="k_resolve k_connect_di" bash profiles/fetch_ab.sh
cd "$(dirname "$0")/.." || exit 1
SCENE=${SCENE:-1}; ALGO=${ALGO:-vcm}; RES=${RES:-2048}; ITER=${ITER:-6}; WARM=${WARM:-2}
export TMPDIR=/tmp
for w in default base; do
  exe=smallvcm_amd/host/vcm_render; [ $w = base ] && exe=profiles/ab_base/host/vcm_render
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/fa_${w}_$c; (cd /tmp && rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/fa_${w}_$c -- $OLDPWD/$exe -s $SCENE -a $ALGO -i $ITER --warmup $WARM --res $RES $RES --json > /dev/null 2>&1)
  done
done
python3 - <<'PY'
import csv, glob, collections, os
want = os.environ.get('KERNELS', '').split()
def load(w, c):
    f = glob.glob('/tmp/fa_%s_%s/**/*counter_collection.csv' % (w, c), recursive=True)[0]
    v = collections.defaultdict(float); n = collections.defaultdict(int); t = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('vcm::', '')
        v[k] += float(r['Counter_Value']); n[k] += 1
        t[k] += float(r['End_Timestamp']) - float(r['Start_Timestamp'])
    return v, n, t
print('%-40s %9s %9s %9s | %9s %9s %9s   (per launch: us, FETCH_SIZE KB-units x1e-6, WRITE_SIZE likewise; raw, uncorrected)' % ('kernel', 'us', 'fetch', 'write', 'base us', 'fetch', 'write'))
d = {w: {c: load(w, c) for c in ('FETCH_SIZE', 'WRITE_SIZE')} for w in ('default', 'base')}
for k in sorted(d['default']['FETCH_SIZE'][0], key=lambda k: -d['default']['FETCH_SIZE'][2][k]):
    if want and not any(x in k for x in want): continue
    row = []
    for w in ('default', 'base'):
        f, n, t = d[w]['FETCH_SIZE']; wv, wn, wt = d[w]['WRITE_SIZE']
        if n.get(k, 0) == 0: row += [0, 0, 0]; continue
        row += [t[k] / n[k] / 1e3, f[k] / n[k] / 1e6, wv.get(k, 0) / max(wn.get(k, 1), 1) / 1e6]
    if row[0] > 15: print('%-40s %9.1f %9.3f %9.3f | %9.1f %9.3f %9.3f' % ((k[:40],) + tuple(row)))
PY
