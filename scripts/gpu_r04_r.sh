#!/bin/bash
# round 4, GPU call R: randomised parity hunts and the soak test on the final tree (new seeds)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04r; mkdir -p $O
cd $R
for seed in 41 42 43; do
  timeout 600 python scripts/fuzz_parity.py 96 $seed > $O/fuzz_parity_$seed.log 2>&1; echo "fuzz_parity seed $seed rc $?"; tail -2 $O/fuzz_parity_$seed.log
done
for seed in 41 42; do
  timeout 600 python scripts/fuzz_engine.py 24 $seed > $O/fuzz_engine_$seed.log 2>&1; echo "fuzz_engine seed $seed rc $?"; tail -2 $O/fuzz_engine_$seed.log
done
timeout 900 python scripts/soak.py 120 > $O/soak.log 2>&1; echo "soak rc $?"; tail -3 $O/soak.log
