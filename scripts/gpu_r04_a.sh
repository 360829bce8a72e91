#!/bin/bash
# round 4, GPU call A: the whole -m gpu suite, per-iteration probes with the team search on / off, the bench line with the driver's
# arguments, the FETCH_SIZE calibration
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04a; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -5 $O/pytest.log
for wl in c4_corridor_1m c2_cylinder_100k c3_pk01_200k; do
  timeout 200 python scripts/run_probe.py $wl > $O/probe_${wl}_team.log 2>&1
  timeout 200 python scripts/run_probe.py $wl team_search=0 > $O/probe_${wl}_noteam.log 2>&1
  tail -2 $O/probe_${wl}_team.log; tail -1 $O/probe_${wl}_noteam.log
done
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err; echo "bench rc $?"; cut -c1-600 $O/bench_steps20.json
timeout 600 bash scripts/microbench/fetch_calib.sh r04 > $O/fetch_calib.log 2>&1; tail -30 $O/fetch_calib.log
