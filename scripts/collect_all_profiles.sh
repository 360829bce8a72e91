#!/bin/bash
# GPU box: rocprofv3 kernel trace + PMC passes for the headline workload and the C2 / C5 configs, then the summaries.
cd "$(dirname "$0")/.."
TAG=${1:-r02}
for WL in c4_corridor_1m c2_cylinder_100k c5_montecarlo_5000; do
  if [ $WL = c5_montecarlo_5000 ]; then export STEPS=1 WARMUP=1 REPEATS=3; else unset STEPS WARMUP REPEATS; fi
  scripts/collect_profiles.sh $TAG $WL > gpurun_out/collect_${TAG}_${WL}.log 2>&1
  python scripts/summarize_profiles.py $TAG $WL > gpurun_out/summary_${TAG}_${WL}.md 2>&1
  tail -25 gpurun_out/summary_${TAG}_${WL}.md
done
mkdir -p gpurun_out/profiles_${TAG}; cp profiles/${TAG}_* gpurun_out/profiles_${TAG}/
