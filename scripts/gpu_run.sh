#!/bin/bash
# One parameterised GPU-box runner (replaces the per-call gpu_rNN_x.sh launchers): runs the named steps in order, everything it writes
# goes to gpurun_out/<tag>/.   usage: scripts/gpu_run.sh <tag> step [step ...]
#   steps: tests[:pytest-args]   pytest -m gpu (default: the whole suite)
#          bench[:bench-args]    python bench.py (default: the driver's --steps 20 --warmup 5) -> <tag>/bench.json
#          profiles[:round-tag]  scripts/collect_all_profiles.sh <round-tag> (kernel trace + PMC passes + summaries, copied to <tag>/)
#          py:<script> [args]    python scripts/<script> ...  -> <tag>/<script>.txt
cd "$(dirname "$0")/.."
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
for STEP in "$@"; do
  KIND=${STEP%%:*}; ARG=""; [ "$STEP" != "$KIND" ] && ARG=${STEP#*:}
  echo "=== $STEP ($(date +%T))"
  case $KIND in
    tests)    timeout 2400 python -m pytest -m gpu -x -q ${ARG:-tests} > $O/tests.log 2>&1; echo "rc=$?"; tail -5 $O/tests.log ;;
    bench)    timeout 1200 python bench.py ${ARG:---steps 20 --warmup 5} > $O/bench.json 2> $O/bench.err; echo "rc=$?"; python scripts/print_bench.py $O/bench.json 2>/dev/null | head -60 ;;
    profiles) timeout 3000 scripts/collect_all_profiles.sh ${ARG:-r06} > $O/profiles.log 2>&1; echo "rc=$?"; tail -30 $O/profiles.log ;;
    py)       S=${ARG%% *}; timeout 1800 python scripts/$ARG > $O/${S%.py}.txt 2>&1; echo "rc=$?"; tail -40 $O/${S%.py}.txt ;;
    *)        echo "unknown step $STEP" ;;
  esac
done
