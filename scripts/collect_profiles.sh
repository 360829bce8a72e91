#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + HBM counters of the bench command.
# Usage: scripts/collect_profiles.sh <round-tag> [workload]
set -u
TAG=${1:-r02}; WL=${2:-c4_corridor_1m}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/prof_${TAG}_${WL}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps ${STEPS:-50} --warmup ${WARMUP:-50} --repeats ${REPEATS:-4} --min-seconds 0 --no-cpu-baseline --no-configs --no-regimes --concurrent-pairs 0 --workload $WL"   # the timed one-pair region only
$BENCH > $O/bench_plain.json 2> $O/bench_plain.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $BENCH > $O/trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $BENCH > $O/fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- $BENCH > $O/write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d $O/sq1 -- $BENCH > $O/sq1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sq2 -- $BENCH > $O/sq2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/tcc -- $BENCH > $O/tcc.log 2>&1
find $O -name "*.csv" | head -40
cat $O/bench_plain.json | cut -c1-400
