#!/bin/bash
# r05y: k_merge_walk_t against k_merge_walk at 2048^2 through the C++ host (same bits? how fast?)
cd smallvcm_amd/host
R="./vcm_render -s 1 -a vcm -i 12 --warmup 3 --res 2048 2048 --json"
p() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['Mpaths_s'], d.get('mean_stats',{}).get('msMergeKernel'), d['image_mean'])"; }
for rep in 1; do
  echo "default-$rep"; timeout 20 $R -o /tmp/a.pfm | p
  echo "transposed-$rep"; SMALLVCM_AMD_MERGE_DRAIN=transposed timeout 20 $R -o /tmp/b.pfm | p
done
cmp /tmp/a.pfm /tmp/b.pfm && echo "TRANSPOSED_SAME_BITS_2048" || echo "TRANSPOSED_DIFFERS_2048"
