#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05l; mkdir -p $O
cd $R
for v in 0 1 0 1 0 1; do
  timeout 300 python bench.py --workload c5_montecarlo_5000 --steps 1 --warmup 1 --repeats 5 --no-configs --no-cpu-baseline --concurrent-pairs 0 --opt pose_zero_copy=$v > $O/c5_$v.json 2>$O/c5_$v.err
  python -c "import json,sys; j=json.load(open('$O/c5_$v.json')); print('c5 pose_zero_copy=$v', round(j['value']), 'it/s', round(j['ms_per_step'],2), 'kernel', round(j['roofline']['kernel_us_avg'],1))"
done
timeout 200 python -m pytest tests/test_gpu_round5.py -q -x -k rccl 2>&1 | tail -2
