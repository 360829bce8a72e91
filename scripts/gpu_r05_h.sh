#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05h; mkdir -p $O
cd $R
for rep in 1 2; do
for wl in c1_fixture_7562 c2_cylinder_100k; do
  for v in "team_pass=0" "team_pass=1"; do
    echo "$wl $v: $(timeout 300 python scripts/run_probe.py $wl $v 2>&1 | tail -4 | tr '\n' ' ')"
  done
done
done
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
