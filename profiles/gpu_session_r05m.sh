#!/bin/bash
# r05m: where the wave time of the big kernels goes (region-clock variant), 2048^2 and 512^2
tag=${1:-r05m}
mkdir -p gpurun_out
export SMALLVCM_AMD_LIB=$PWD/smallvcm_amd/csrc/libsmallvcm_amd_rc.so
( timeout 120 python profiles/tools/region_clock.py 1 2048 vcm
  timeout 120 python profiles/tools/region_clock.py 1 512 vcm ) > gpurun_out/${tag}_region_clock.txt 2>&1
cat gpurun_out/${tag}_region_clock.txt
