#!/bin/bash
# A/B of one backend option on the Monte-Carlo experiment (bench.py --workload c5_montecarlo_5000, three experiments per line) on ONE box.
# usage: scripts/ab_mc.sh ["opt=a" "opt=b" ...]      (default: the one-wave blocks of the batches off / on)
cd "$(dirname "$0")/.."
[ $# -eq 0 ] && set -- "one_wave_batches=0" "one_wave_batches=1"
for r in 1 2 3; do
  for o in "$@"; do
    python bench.py --workload c5_montecarlo_5000 --steps 3 --warmup 1 --min-seconds 0 --no-cpu-baseline --no-configs --no-regimes --concurrent-pairs 0 --opt $o 2>/dev/null | tail -1 |
      python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$o', round(d['value'], 1), 'it/s', round(d['ms_per_step'], 2), 'ms per experiment')"
  done
done
