#!/bin/bash
# GPU box: the concurrent-pairs leg of bench.py (two contexts, two host threads, gated launches on two streams) N times in fresh processes;
# prints how many runs failed and with what.   usage: scripts/concurrent_soak.sh [N]
cd "$(dirname "$0")/.."
N=${1:-10}; bad=0
for i in $(seq 1 $N); do
  out=$(timeout 300 python bench.py --no-configs --no-regimes --no-cpu-baseline --min-seconds 1 --steps 50 --warmup 50 2>&1)
  rc=$?
  if [ $rc -ne 0 ] || echo "$out" | grep -q '"concurrent_pairs": {"error"'; then
    bad=$((bad + 1)); echo "run $i: rc=$rc"; echo "$out" | grep -v amdgpu.ids | tail -3 | cut -c1-400
  else
    echo "run $i ok: $(echo "$out" | python -c 'import sys, json; d = json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(round(d["value"]), round(d["concurrent_pairs"]["value"]))')"
  fi
done
echo "concurrent soak: $bad failures in $N runs"
