#!/bin/bash
# SQ counter passes of the headline run (each --pmc set in its own run, with --kernel-trace only)
set -u
TAG=${1:-r02e}
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/${TAG}
B="python bench.py --no-cpu-baseline --no-configs --no-traffic --steps 6 --warmup 2"
rocprofv3 -L > ${O}_counters.txt 2>&1
i=0
while read -r set; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/${TAG}_pmc$i -- $B > ${O}_pmc$i.log 2>&1
  echo "pass $i rc=$? : $set" >> ${O}_box.log
done <<'SETS'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC
SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_INSTS_FLAT SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_IFETCH
SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32
SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_IFETCH_LEVEL SQ_INSTS_VALU_FMA_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_EXP_GDS
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum
SETS
ls gpurun_out/${TAG}_pmc*/*/ 2>/dev/null | head -40
cat ${O}_box.log
