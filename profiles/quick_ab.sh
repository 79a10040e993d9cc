#!/bin/bash
# The fast A/B loop: variant builds of the library against the default through the C++ host (smallvcm_amd/host/vcm_render:
# no Python, no torch import -- a gpurun call of this costs ~11 s of GPU budget instead of ~2 min).
#
#   make -C smallvcm_amd/csrc variant NAME=x EXTRA=-DSOMETHING      (here, no GPU needed; builds libsmallvcm_amd_x.so)
#   gpurun --timeout 60 -- 'VARIANTS="x y" REPS=2 bash profiles/quick_ab.sh > gpurun_out/ab.txt 2>&1; cat gpurun_out/ab.txt'
#
# Per variant: Mpaths/s of scene $SCENE, algorithm $ALGO at $RES^2 over $ITER iterations after $WARM warm-up iterations,
# and whether the frame is the default build's bit for bit (cmp of the raw fp32 PFM).  ENVS="A=1 B=2" adds runs of the
# DEFAULT build with one environment switch each.  12-iteration runs scatter by +-2 %: repeat (REPS) before believing 1 %.
cd "$(dirname "$0")/../smallvcm_amd/host" || exit 1
SCENE=${SCENE:-1}; ALGO=${ALGO:-vcm}; RES=${RES:-2048}; ITER=${ITER:-12}; WARM=${WARM:-3}; REPS=${REPS:-1}
R="./vcm_render -s $SCENE -a $ALGO -i $ITER --warmup $WARM --res $RES $RES --json"
p() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%8.1f Mpaths/s  mean %s' % (d['Mpaths_s'], d['image_mean']))"; }
for rep in $(seq 1 $REPS); do
  printf "%-28s" "default"; timeout 30 $R -o /tmp/qab_a.pfm | p
  for v in ${VARIANTS:-}; do
    printf "%-28s" "$v"; LD_PRELOAD=$PWD/../csrc/libsmallvcm_amd_$v.so timeout 30 $R -o /tmp/qab_b.pfm | p
    cmp -s /tmp/qab_a.pfm /tmp/qab_b.pfm && echo "    same bits as the default" || echo "    DIFFERS from the default"
  done
  for e in ${ENVS:-}; do
    printf "%-28s" "$e"; env "$e" timeout 30 $R -o /tmp/qab_b.pfm | p
    cmp -s /tmp/qab_a.pfm /tmp/qab_b.pfm && echo "    same bits as the default" || echo "    DIFFERS from the default"
  done
done
