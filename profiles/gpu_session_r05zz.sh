#!/bin/bash
# r05zz: k_merge_walk draining when K lanes are full (VCM_K4_FULL = 4 / 8 / 12 / 16 / 24) against the default (1), C++ host
cd smallvcm_amd/host
R="./vcm_render -s 1 -a vcm -i 12 --warmup 3 --res 2048 2048 --json"
p() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['Mpaths_s'], d['image_mean'])"; }
echo "default"; timeout 12 $R -o /tmp/a.pfm | p
for v in ${VARIANTS:-f4 f8 f12 f16}; do
  echo "$v"; LD_PRELOAD=$PWD/../csrc/libsmallvcm_amd_$v.so timeout 12 $R -o /tmp/b.pfm | p
  cmp /tmp/a.pfm /tmp/b.pfm && echo "SAME_BITS_$v" || echo "DIFFERS_$v"
done
echo "default again"; timeout 12 $R -o /tmp/a.pfm | p
