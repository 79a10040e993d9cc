#!/bin/bash
# The ONE script behind every gpurun call of round 5 (rounds 1-4 kept a gpu_session_<tag>.sh per call: 55 of them; their
# commands are in git history and in the logs they produced).  Steps run in the order given; every step writes
# gpurun_out/<tag>_<step>.*, so a call that is cut off still leaves what it finished.
#
#   gpurun --timeout 900 -- 'bash profiles/session.sh r07a ab512 ab1024 timeline512 bench_nocpu'
#
# steps
#   pytest            the whole GPU suite                               -> <tag>_pytest.log
#   pytest:<expr>     ... restricted with -k <expr>
#   bench             python bench.py --steps 20 --warmup 5 (the driver's command; CPU legs included: minutes)
#   bench_nocpu       ... --cpu-baseline none
#   bench_record      ... --cpu-baseline none --record-counters <tag>  (keeps profiles/<tag>_counters_<config>.json)
#   collect           profiles/collect.sh <tag> on the headline (kernel stats, FETCH / WRITE, SQ passes)
#   collect:<cfg>     ... on C1 | C2 | C3
#   kstats[:res[:algo[:scene]]]   profiles/kstats.sh: average us per launch of the big kernels under --kernel-trace (VARIANTS, ENVS as for ab)
#   mem[:<args>]      profiles/collect_mem.sh: TA / TD / TCP / TCC counter passes (default C4; args = vcm_render arguments)
#   timeline<res>     profiles/tools/timeline.py of scene 1 vcm at <res>^2 (timeline1024s3: scene 3)
#   ab<res>[:algo[:scene]]   profiles/quick_ab.sh (variants from $VARIANTS, switches from $ENVS, $REPS repetitions, $ITER iterations)
#   farm:<ranks>:<shards>:<inflight>[:res]   vcm_render --gpus <ranks> --collectives threads on this one GPU
#   dropin<res>       smallvcm_amd/dropin/dropin_rate[_addcolor]: the reference's renderer interface over the drop-in (host Framebuffer
#                     refreshed after every RunIteration) against the C-ABI alone
#   sh:<command>      anything else, verbatim
set -u
TAG=${1:-rXX}; shift
export TMPDIR=/tmp
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
O=gpurun_out/${TAG}
for step in "$@"; do
  echo "=== $step ($(date +%T))"
  case "$step" in
    pytest)        timeout 1500 python -m pytest tests -m gpu -x -q > ${O}_pytest.log 2>&1; tail -3 ${O}_pytest.log ;;
    pytest:*)      timeout 1200 python -m pytest tests -m gpu -x -q -k "${step#pytest:}" > ${O}_pytest_k.log 2>&1; tail -3 ${O}_pytest_k.log ;;
    bench)         timeout 1500 python bench.py --steps 20 --warmup 5 > ${O}_bench.log 2> ${O}_bench.err; tail -c 4200 ${O}_bench.log; cp -f bench_detail.json ${O}_bench_detail.json 2>/dev/null ;;
    bench_nocpu)   timeout 900 python bench.py --steps 20 --warmup 5 --cpu-baseline none > ${O}_bench_nocpu.log 2> ${O}_bench_nocpu.err; tail -c 4200 ${O}_bench_nocpu.log; tail -5 ${O}_bench_nocpu.err; cp -f bench_detail.json ${O}_bench_nocpu_detail.json 2>/dev/null ;;
    bench_record)  timeout 900 python bench.py --steps 20 --warmup 5 --cpu-baseline none --record-counters ${TAG} > ${O}_bench_record.log 2> ${O}_bench_record.err; tail -c 4200 ${O}_bench_record.log; cp -f bench_detail.json ${O}_bench_record_detail.json 2>/dev/null; cp -f profiles/${TAG}_counters_*.json gpurun_out/ 2>/dev/null ;;
    collect)       BENCH_ARGS="--steps 20 --warmup 5" timeout 900 bash profiles/collect.sh ${TAG} > ${O}_collect.log 2>&1; tail -3 ${O}_collect.log ;;
    collect:C1)    BENCH_ARGS="--steps 20 --warmup 5 --scene 1 --algo vcm --res 512" timeout 600 bash profiles/collect.sh ${TAG}_C1 > ${O}_collect_C1.log 2>&1; tail -2 ${O}_collect_C1.log ;;
    collect:C2)    BENCH_ARGS="--steps 20 --warmup 5 --scene 3 --algo vcm --res 1024" timeout 600 bash profiles/collect.sh ${TAG}_C2 > ${O}_collect_C2.log 2>&1; tail -2 ${O}_collect_C2.log ;;
    collect:C3)    BENCH_ARGS="--steps 20 --warmup 5 --scene 1 --algo bpm --res 2048" timeout 600 bash profiles/collect.sh ${TAG}_C3 > ${O}_collect_C3.log 2>&1; tail -2 ${O}_collect_C3.log ;;
    kstats*)       spec=${step#kstats}; spec=${spec#:}; IFS=: read -r r a sc <<< "$spec"
                   RES=${r:-2048} ALGO=${a:-vcm} SCENE=${sc:-1} timeout 900 bash profiles/kstats.sh > ${O}_kstats_${r:-2048}_${a:-vcm}.txt 2>&1; cat ${O}_kstats_${r:-2048}_${a:-vcm}.txt ;;
    mem)           timeout 1500 bash profiles/collect_mem.sh ${TAG} > ${O}_mem.log 2>&1; tail -150 ${O}_mem.log ;;
    mem:*)         timeout 1500 bash profiles/collect_mem.sh ${TAG} ${step#mem:} > ${O}_mem.log 2>&1; tail -150 ${O}_mem.log ;;
    timeline1024s3) timeout 300 python profiles/tools/timeline.py ${TAG} --res 1024 --scene 3 > /dev/null 2>&1; head -60 ${O}_timeline1024.txt ;;
    timeline*)     r=${step#timeline}; timeout 300 python profiles/tools/timeline.py ${TAG} --res $r > /dev/null 2>&1; head -60 ${O}_timeline${r}.txt ;;
    ab*)           spec=${step#ab}; IFS=: read -r r a s <<< "$spec"
                   RES=$r ALGO=${a:-vcm} SCENE=${s:-1} ITER=${ITER:-40} WARM=${WARM:-5} REPS=${REPS:-2} timeout 600 bash profiles/quick_ab.sh > ${O}_ab_${r}_${a:-vcm}_s${s:-1}.txt 2>&1
                   cat ${O}_ab_${r}_${a:-vcm}_s${s:-1}.txt ;;
    farm:*)        IFS=: read -r _ n sh fl r <<< "$step"; r=${r:-2048}
                   ( cd smallvcm_amd/host && timeout 300 ./vcm_render -s 1 -a vcm -i ${ITER:-20} --warmup ${WARM:-5} --res $r $r --gpus $n --shards $sh --inflight $fl --collectives threads --same-window --json ) > ${O}_farm_${n}_${sh}_${fl}_${r}.txt 2>&1
                   tail -c 1500 ${O}_farm_${n}_${sh}_${fl}_${r}.txt ;;
    dropin*)       r=${step#dropin}; ( timeout 300 smallvcm_amd/dropin/dropin_rate $r ${ITER:-20} ${WARM:-5}; timeout 300 smallvcm_amd/dropin/dropin_rate_addcolor $r ${ITER:-20} ${WARM:-5}; timeout 300 smallvcm_amd/dropin/dropin_rate $r ${ITER:-20} ${WARM:-5} 1 vcm 2; timeout 300 smallvcm_amd/dropin/dropin_rate $r ${ITER:-20} ${WARM:-5} 1 vcm 4 ) > ${O}_dropin${r}.txt 2>&1; cat ${O}_dropin${r}.txt ;;
    sh:*)          timeout 900 bash -c "${step#sh:}" > ${O}_sh.log 2>&1; tail -20 ${O}_sh.log ;;
    *)             echo "unknown step $step" ;;
  esac
done
echo "=== done ($(date +%T))"
