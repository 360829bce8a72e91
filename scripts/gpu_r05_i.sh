#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05i; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for wl in c1_fixture_7562 c2_cylinder_100k c4_corridor_1m; do
  echo "$wl default: $(timeout 300 python scripts/run_probe.py $wl 2>&1 | tail -4 | tr '\n' ' ')"
done
timeout 300 python scripts/reg_probe.py 2>&1 | grep -E "registration|per-iter|^pass"
timeout 300 python bench.py --steps 20 --warmup 5 --repeats 60 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python scripts/print_bench.py $O/bench.json 2>/dev/null | head -40
