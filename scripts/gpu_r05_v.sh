#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05v; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q -k "trials or montecarlo" 2>&1 | tail -2
for v in 0 1 0 1 0 1; do
  if [ $v = 0 ]; then export DCREG_TRIALS_NO_LPT=1; else unset DCREG_TRIALS_NO_LPT; fi
  timeout 300 python bench.py --workload c5_montecarlo_5000 --steps 1 --warmup 1 --repeats 5 --no-configs --no-cpu-baseline --concurrent-pairs 0 > $O/c5_$v.json 2>$O/c5_$v.err
  python -c "import json,sys; j=json.loads(open('$O/c5_$v.json').read().strip().split('\n')[-1]); print('c5 lpt=$v', round(j['value']), 'it/s', round(j['ms_per_step'],2), 'kernel', round(j['roofline']['kernel_us_avg'],1))"
done
