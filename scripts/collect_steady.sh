#!/bin/bash
# GPU box: the settled launches of a workload (scripts/steady_probe.py: 50 iterations to converge, then N more) under rocprofv3:
# kernel trace + HBM and SQ counters; prints the figures of the LAST N k_lin dispatches of every pass.
# usage: scripts/collect_steady.sh <round-tag> [workload] [N]
TAG=${1:-r03}; WL=${2:-c4_corridor_1m}; N=${3:-100}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/prof_${TAG}_steady_${WL}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/scripts/steady_probe.py $WL $N"
$CMD > $O/plain.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- $CMD > $O/trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $CMD > $O/fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- $CMD > $O/write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d $O/sq1 -- $CMD > $O/sq1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/tcc -- $CMD > $O/tcc.log 2>&1
python $R/scripts/summarize_steady.py "$O" "$N" "$TAG" "$WL"
