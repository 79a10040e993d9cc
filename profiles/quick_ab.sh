#!/bin/bash
# The fast A/B loop: variant builds of the library against the default through the C++ host (smallvcm_amd/host/vcm_render:
# no Python, no torch import -- a gpurun call of this costs ~11 s of GPU budget instead of ~2 min).
#
#   bash profiles/make_variant.sh x "-DSOMETHING"      (here, no GPU needed; builds profiles/ab_x/{csrc,host})
#   gpurun --timeout 60 -- 'VARIANTS="x y" REPS=2 bash profiles/quick_ab.sh > gpurun_out/ab.txt 2>&1; cat gpurun_out/ab.txt'
#
# Per variant: Mpaths/s of scene $SCENE, algorithm $ALGO at $RES^2 over $ITER iterations after $WARM warm-up iterations,
# and whether the frame is the default build's bit for bit (cmp of the raw fp32 PFM).  ENVS="A=1 B=2" adds runs of the
# DEFAULT build with one environment switch each.  BASE=1 adds the row "base": another revision (profiles/make_ab_base.sh).  12-iteration runs scatter by +-2 %: repeat (REPS) before believing 1 %.
cd "$(dirname "$0")/../smallvcm_amd/host" || exit 1
SCENE=${SCENE:-1}; ALGO=${ALGO:-vcm}; RES=${RES:-2048}; ITER=${ITER:-12}; WARM=${WARM:-3}; REPS=${REPS:-1}
R="./vcm_render -s $SCENE -a $ALGO -i $ITER --warmup $WARM --res $RES $RES --json"
p() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('last_iteration_kernel_ms',{}); print('%8.1f Mpaths/s  K1 %.2f K3 %.2f K3b %.2f K4 %.2f grid %.2f  [%s]' % (d['Mpaths_s'], k.get('light',0), k.get('camera',0), k.get('connect_di',0), k.get('merge',0), k.get('grid_side',0), d.get('library','?')))"; }
RB="../../profiles/ab_base/host/vcm_render -s $SCENE -a $ALGO -i $ITER --warmup $WARM --res $RES $RES --json"
for rep in $(seq 1 $REPS); do
  printf "%-28s" "default"; timeout 60 $R -o /tmp/qab_a.pfm | p
  if [ -n "${BASE:-}" ]; then   # another REVISION: its own host + library (profiles/make_ab_base.sh)
    printf "%-28s" "base $(cat ../../profiles/ab_base/REV)"; timeout 30 $RB -o /tmp/qab_b.pfm | p
    cmp -s /tmp/qab_a.pfm /tmp/qab_b.pfm && echo "    same bits as the default" || echo "    DIFFERS from the default"
  fi
  for v in ${VARIANTS:-}; do   # profiles/make_variant.sh: a complete pair (library + host) per variant
    printf "%-28s" "$v"; timeout 30 ../../profiles/ab_$v/host/vcm_render -s $SCENE -a $ALGO -i $ITER --warmup $WARM --res $RES $RES --json -o /tmp/qab_b.pfm | p
    cmp -s /tmp/qab_a.pfm /tmp/qab_b.pfm && echo "    same bits as the default" || echo "    DIFFERS from the default"
  done
  for e in ${ENVS:-}; do
    printf "%-44s" "$e"; env ${e//,/ } timeout 60 $R -o /tmp/qab_b.pfm | p
    cmp -s /tmp/qab_a.pfm /tmp/qab_b.pfm && echo "    same bits as the default" || echo "    DIFFERS from the default"
  done
done
