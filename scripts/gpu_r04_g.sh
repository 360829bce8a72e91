#!/bin/bash
# round 4, GPU call G: where the slow repeat of the bench loop comes from; probes with events on every launch
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04g; mkdir -p $O
cd $R
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-regimes --concurrent-pairs 0 > $O/bench_$i.json 2> $O/bench_$i.err
python -c "import json; j=json.loads(open('$O/bench_$i.json').read().strip().split('\n')[-1]); print('run $i', j['value'], j['ms_per_step'], j['ms_per_step_median'], j['slowest_repeats'], j['roofline']['kernel_us_avg'])"
done
timeout 300 python bench.py --steps 50 --warmup 50 --no-cpu-baseline --no-configs --no-regimes --concurrent-pairs 0 > $O/bench_50.json 2> $O/bench_50.err
python -c "import json; j=json.loads(open('$O/bench_50.json').read().strip().split('\n')[-1]); print('steps50', j['value'], j['ms_per_step'], j['ms_per_step_median'], j['slowest_repeats'], j['roofline']['kernel_us_avg'])"
timeout 200 python scripts/run_probe.py c4_corridor_1m > $O/probe_plain.log 2>&1; tail -2 $O/probe_plain.log
timeout 200 python scripts/run_probe.py c4_corridor_1m time_kernels=1 > $O/probe_timed.log 2>&1; tail -2 $O/probe_timed.log
