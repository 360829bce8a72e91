#!/bin/bash
# round 4, GPU call K: the final code - suite, the bench line with the driver's arguments and with the default ones, C4 profile, wave phases
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04k; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_line_steps20.json 2> $O/bench_line_steps20.err; echo "bench steps20 rc $?"
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench_line.err; echo "bench default rc $?"
python scripts/print_bench.py $O/bench_line_steps20.json 2>/dev/null | head -40
REPEATS=30 timeout 900 bash scripts/collect_profiles.sh r04 c4_corridor_1m > $O/collect_c4.log 2>&1; tail -3 $O/collect_c4.log
timeout 300 python scripts/wave_phases.py c4_corridor_1m > $O/phases_c4.log 2>&1; tail -4 $O/phases_c4.log
