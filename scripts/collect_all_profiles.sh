#!/bin/bash
# GPU box: rocprofv3 kernel trace + PMC passes for all five BASELINE configs and the settled launches of C4, then the summaries
# (profiles/<tag>_*; summarize_profiles.py refuses a trace whose kernels this tree's library does not export).
cd "$(dirname "$0")/.."
TAG=${1:-r02}
for WL in c4_corridor_1m c2_cylinder_100k c3_pk01_200k c1_fixture_7562 c5_montecarlo_5000; do
  if [ $WL = c5_montecarlo_5000 ]; then export STEPS=1 WARMUP=1 REPEATS=3; else unset STEPS WARMUP REPEATS; fi
  scripts/collect_profiles.sh $TAG $WL > gpurun_out/collect_${TAG}_${WL}.log 2>&1
  python scripts/summarize_profiles.py $TAG $WL > gpurun_out/summary_${TAG}_${WL}.md 2>&1
  tail -25 gpurun_out/summary_${TAG}_${WL}.md
done
scripts/collect_steady.sh $TAG c4_corridor_1m 100 > gpurun_out/collect_${TAG}_steady.log 2>&1; tail -12 gpurun_out/collect_${TAG}_steady.log
mkdir -p gpurun_out/profiles_${TAG}; cp profiles/${TAG}_* gpurun_out/profiles_${TAG}/
# (the raw traces run to hundreds of MB and gpurun merges at most 64 MiB back: the summaries and the rocprofv3 --stats tables are what is kept)
rm -rf gpurun_out/prof_${TAG}_*
