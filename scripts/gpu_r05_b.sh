#!/bin/bash
# round 5, GPU call B: round-5 tests on the new tree; A/B of the sweep's layer batch (1 = one layer at a time as before, 2, 4 = product)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05b; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_round5.py -x -q > $O/tests_round5.log 2>&1; echo "round5 tests rc $?"; tail -3 $O/tests_round5.log
bash scripts/ab_multi.sh "sb1 sb2 product" 2 "c4_corridor_1m c3_pk01_200k" > $O/ab.log 2>&1; grep "per-iteration" $O/ab.log | cut -c1-260
grep "sum" $O/ab.log | sed 's/.*\(c[0-9]_[a-z0-9_]* [a-z0-9]*\):.*sum \([0-9]*\) us.*/\1 \2/'
timeout 300 python bench.py --steps 20 --warmup 5 --repeats 60 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python scripts/print_bench.py $O/bench.json 2>/dev/null | head -40
