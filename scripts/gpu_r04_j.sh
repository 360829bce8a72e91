#!/bin/bash
# round 4, GPU call J: bench at the driver's arguments (final code), rocprofv3 profiles of C4 / C2 / C5 and of the settled C4 launches
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04j; mkdir -p $O
cd $R
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-regimes --concurrent-pairs 0 > $O/bench_quick.json 2> $O/bench_quick.err
python -c "import json; j=json.loads(open('$O/bench_quick.json').read().strip().split('\n')[-1]); print('steps20', j['value'], j['ms_per_step'], j['ms_per_step_median'], j['slowest_repeats'], j['roofline']['kernel_us_avg'])"
timeout 1500 bash scripts/collect_all_profiles.sh r04 > $O/collect_all.log 2>&1; tail -40 $O/collect_all.log
timeout 600 bash scripts/collect_steady.sh r04 c4_corridor_1m 100 > $O/collect_steady.log 2>&1; tail -30 $O/collect_steady.log
ls $R/profiles | grep r04
