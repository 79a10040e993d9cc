#!/bin/bash
# round 4, first GPU session: the whole GPU suite on the glibc-exact detmath + new tests, A/B against round 3's library,
# one bench line through the new N=1 path (C++ farm + rocprofv3 children on the run's window)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
nproc > gpurun_out/r06a_box.log; rocm-smi --showproductname 2>/dev/null | head -8 >> gpurun_out/r06a_box.log
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r06a_pytest.log 2>&1
tail -5 gpurun_out/r06a_pytest.log
VARIANTS="r3" REPS=3 bash profiles/quick_ab.sh > gpurun_out/r06a_ab.txt 2>&1
cat gpurun_out/r06a_ab.txt
SCENE=3 RES=1024 VARIANTS="r3" REPS=1 bash profiles/quick_ab.sh > gpurun_out/r06a_ab_s3.txt 2>&1
RES=512 ITER=40 VARIANTS="r3" REPS=1 bash profiles/quick_ab.sh >> gpurun_out/r06a_ab_s3.txt 2>&1
cat gpurun_out/r06a_ab_s3.txt
timeout 600 python bench.py --cpu-baseline none --no-configs > gpurun_out/r06a_bench.log 2> gpurun_out/r06a_bench.err
tail -c 3000 gpurun_out/r06a_bench.log; tail -5 gpurun_out/r06a_bench.err
