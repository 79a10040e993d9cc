#!/bin/bash
# Regenerates the profiles of a round ON THE GPU BOX (run through gpurun from the repo root):
#   bash profiles/collect.sh r02a            (BENCH_ARGS="--res 512" etc. selects another configuration)
# writes gpurun_out/<tag>_*; copy the summaries into profiles/ afterwards.
set -u
TAG=${1:-rXX}
export TMPDIR=/tmp
mkdir -p gpurun_out
CMD="python bench.py --no-cpu-baseline --no-configs --no-traffic --no-cross-check --no-dropin ${BENCH_ARGS:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats -- $CMD > gpurun_out/${TAG}_profiled_run.log 2>&1
# counters in their own runs, never combined with trace domains other than --kernel-trace
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${TAG}_fetch -- $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/${TAG}_write -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU \
          --kernel-trace --output-format csv -d gpurun_out/${TAG}_sq1 -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 \
          --kernel-trace --output-format csv -d gpurun_out/${TAG}_sq2 -- $CMD > /dev/null 2>&1
# what the waves wait for: scalar / vector memory latency (LEVEL / INSTS), instruction fetch, issue stalls
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_IFETCH SQ_IFETCH_LEVEL \
          --kernel-trace --output-format csv -d gpurun_out/${TAG}_sq3 -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_TRANS_F32 \
          --kernel-trace --output-format csv -d gpurun_out/${TAG}_sq4 -- $CMD > /dev/null 2>&1
ls gpurun_out/${TAG}_*
