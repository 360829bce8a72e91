#!/bin/bash
# round 5: rocprofv3 kernel trace + PMC passes (separate runs) of every BASELINE config's bench command and of the settled C4 launches; summaries -> profiles/r05_*
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
TAG=r05
for WL in ${1:-c4_corridor_1m c2_cylinder_100k c3_pk01_200k c1_fixture_7562 c5_montecarlo_5000}; do
  if [ $WL = c5_montecarlo_5000 ]; then export STEPS=1 WARMUP=1 REPEATS=3; else unset STEPS WARMUP REPEATS; fi
  scripts/collect_profiles.sh $TAG $WL > gpurun_out/collect_${TAG}_${WL}.log 2>&1
  python scripts/summarize_profiles.py $TAG $WL > gpurun_out/summary_${TAG}_${WL}.md 2>&1
  grep -E "linearisation|Un-profiled|HBM-side|FETCH_SIZE .* KB" gpurun_out/summary_${TAG}_${WL}.md | head -6
done
if [ -z "$1" ]; then scripts/collect_steady.sh $TAG c4_corridor_1m 100 2>&1 | tail -12; fi
mkdir -p gpurun_out/profiles_${TAG}; cp profiles/${TAG}_* gpurun_out/profiles_${TAG}/
