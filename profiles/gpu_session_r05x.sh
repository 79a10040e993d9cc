#!/bin/bash
# r05x: does k_merge_walk_t (transposed drain, experimental) give the default kernel's frame?  C++ host only (no Python
# start-up): 75 s of GPU budget were left.
cd smallvcm_amd/host
R="./vcm_render -s 1 -a vcm -i 4 --res 512 512 --json"
timeout 25 $R -o /tmp/a.pfm | tail -1 | cut -c1-200
SMALLVCM_AMD_MERGE_DRAIN=transposed timeout 25 $R -o /tmp/b.pfm | tail -1 | cut -c1-200
cmp /tmp/a.pfm /tmp/b.pfm && echo "TRANSPOSED_SAME_BITS_VCM" || echo "TRANSPOSED_DIFFERS_VCM"
R2="./vcm_render -s 1 -a bpm -i 3 --res 512 512 --json"
timeout 20 $R2 -o /tmp/c.pfm | tail -1 | cut -c1-160
SMALLVCM_AMD_MERGE_DRAIN=transposed timeout 20 $R2 -o /tmp/d.pfm | tail -1 | cut -c1-160
cmp /tmp/c.pfm /tmp/d.pfm && echo "TRANSPOSED_SAME_BITS_BPM" || echo "TRANSPOSED_DIFFERS_BPM"
