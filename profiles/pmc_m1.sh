#!/bin/bash
# SQ counters per kernel for the scene-file configuration M1 (BVH, 1024^2): two --pmc passes, summed per kernel name
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
ROOT=$PWD
LIB=${LIBDIR:-$ROOT/smallvcm_amd}
CMD="$LIB/host/vcm_render --scene-file $ROOT/tests/scenes/bumpy_room.vcmscene -a vcm -i 6 --warmup 2 --res 1024 1024 --json"
rm -rf /tmp/pm1a /tmp/pm1b
(cd /tmp && rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pm1a -- $CMD > /dev/null 2>&1)
(cd /tmp && rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d /tmp/pm1b -- $CMD > /dev/null 2>&1)
python3 - <<'PY'
import csv, glob, collections
tab = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(float); calls = collections.defaultdict(int)
for d in ('/tmp/pm1a', '/tmp/pm1b'):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('vcm::', '')
        tab[n][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] in ('SQ_WAVES',):
            calls[n] += 1
            dur[n] += (float(r['End_Timestamp']) - float(r['Start_Timestamp'])) if 'End_Timestamp' in r else 0
print('%-40s %6s %8s %8s %6s %6s %7s %7s %7s %7s' % ('kernel (8 iterations)', 'ms/it', 'VALU M', 'SALU M', 'lanes', 'valu%', 'wait%', 'VMEM M', 'LDS M', 'vmlvl'))
for n, c in sorted(tab.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0)):
    if c.get('SQ_INSTS_VALU', 0) < 1e6: continue
    it = 8.0
    wc = c['SQ_WAVE_CYCLES']
    print('%-40s %6.3f %8.1f %8.1f %6.2f %6.1f %7.1f %7.1f %7.1f %7.1f' % (n[:40], dur[n] / it / 1e6, c['SQ_INSTS_VALU'] / it / 1e6, c['SQ_INSTS_SALU'] / it / 1e6,
          c['SQ_THREAD_CYCLES_VALU'] / max(c['SQ_ACTIVE_INST_VALU'], 1) / 64, 100 * 4 * c['SQ_INSTS_VALU'] / wc, 100 * 4 * c['SQ_WAIT_INST_ANY'] / wc,
          c['SQ_INSTS_VMEM_RD'] / it / 1e6, c['SQ_INSTS_LDS'] / it / 1e6, c['SQ_INST_LEVEL_VMEM'] / max(c['SQ_BUSY_CYCLES'],1)))
PY
