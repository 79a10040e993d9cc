#!/bin/bash
# the mesh scene M1 through the default build and the cross-revision baseline (profiles/make_ab_base.sh)
cd "$(dirname "$0")/.." || exit 1
ITER=${ITER:-40}; WARM=${WARM:-5}; REPS=${REPS:-3}
A="--scene-file tests/scenes/bumpy_room.vcmscene -a ${ALGO:-vcm} -i $ITER --warmup $WARM --res ${RES:-1024} ${RES:-1024} --json"
p() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('last_iteration_kernel_ms',{}); print('%8.1f Mpaths/s  K1 %.2f K3 %.2f K3b %.2f K4 %.2f  [%s]' % (d['Mpaths_s'], k.get('light',0), k.get('camera',0), k.get('connect_di',0), k.get('merge',0), d.get('library','?')))"; }
for rep in $(seq 1 $REPS); do
  printf "%-22s" default; smallvcm_amd/host/vcm_render $A -o /tmp/m1_a.pfm 2>/dev/null | p
  printf "%-22s" "base $(cat profiles/ab_base/REV)"; profiles/ab_base/host/vcm_render $A -o /tmp/m1_b.pfm 2>/dev/null | p
  cmp -s /tmp/m1_a.pfm /tmp/m1_b.pfm && echo "    same bits as the default" || echo "    DIFFERS from the default"
  for e in ${ENVS:-}; do
    printf "%-22s" "$e"; env "$e" smallvcm_amd/host/vcm_render $A -o /tmp/m1_b.pfm 2>/dev/null | p
    cmp -s /tmp/m1_a.pfm /tmp/m1_b.pfm && echo "    same bits as the default" || echo "    DIFFERS from the default"
  done
done
