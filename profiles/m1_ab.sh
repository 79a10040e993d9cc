#!/bin/bash
# the mesh scene M1 (scene file, BVH, VCM 1024^2) through variant pairs (profiles/make_variant.sh): Mpaths/s, bits against the default
cd "$(dirname "$0")/.." || exit 1
ITER=${ITER:-40}; WARM=${WARM:-5}; REPS=${REPS:-2}
A="--scene-file tests/scenes/bumpy_room.vcmscene -a vcm -i $ITER --warmup $WARM --res 1024 1024 --json"
p() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('last_iteration_kernel_ms',{}); print('%8.1f Mpaths/s  K1 %.2f K3 %.2f K3b %.2f K4 %.2f  [%s]' % (d['Mpaths_s'], k.get('light',0), k.get('camera',0), k.get('connect_di',0), k.get('merge',0), d.get('library','?')))"; }
for rep in $(seq 1 $REPS); do
  printf "%-12s" default; smallvcm_amd/host/vcm_render $A -o /tmp/m1_a.pfm 2>/dev/null | p
  for v in ${VARIANTS:-}; do
    printf "%-12s" $v; profiles/ab_$v/host/vcm_render $A -o /tmp/m1_b.pfm 2>/dev/null | p
    cmp -s /tmp/m1_a.pfm /tmp/m1_b.pfm && echo "    same bits as the default" || echo "    DIFFERS from the default"
  done
done
