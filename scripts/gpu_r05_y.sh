#!/bin/bash
# round 5, call Y: more randomised hunts on the final tree (new seeds)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05y; mkdir -p $O
cd $R
for seed in 15 16 17 18; do
  timeout 400 python scripts/fuzz_passes.py 100 $seed > $O/fuzz_passes_$seed.log 2>&1; echo "fuzz_passes seed $seed rc $?"; tail -1 $O/fuzz_passes_$seed.log
done
for seed in 61 62; do
  timeout 500 python scripts/fuzz_engine.py 24 $seed > $O/fuzz_engine_$seed.log 2>&1; echo "fuzz_engine seed $seed rc $?"; tail -2 $O/fuzz_engine_$seed.log
done
