#!/bin/bash
# A/B of one backend option with the DRIVER's bench arguments (--steps 20 --warmup 5: a fence every 20 steps) on ONE box, alternating.
# usage: scripts/ab_steps20.sh ["opt=a" "opt=b" ...]      (default: the one-wave launches off / by the rule)
cd "$(dirname "$0")/.."
[ $# -eq 0 ] && set -- "one_wave=0" "one_wave=1"
for r in 1 2 3; do
  for o in "$@"; do
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-regimes --concurrent-pairs 0 --opt $o 2>/dev/null | tail -1 |
      python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$o', round(d['value'], 1), 'it/s', round(1e3 * d['ms_per_step'], 2), 'us per step')"
  done
done
