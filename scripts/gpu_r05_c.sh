#!/bin/bash
# round 5, GPU call C: the small-frame pass - tests, registration probe with the pass off / by rule / forced, C1 / C2 with and without it, C5 with the pose copy on either stream
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05c; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_round5.py -x -q > $O/tests_round5.log 2>&1; echo "round5 tests rc $?"; tail -5 $O/tests_round5.log
for v in "team_pass=0" "team_pass=1" "team_pass=2"; do
  timeout 300 python scripts/reg_probe.py $v > $O/reg_$v.log 2>&1; echo "reg $v rc $?"; cat $O/reg_$v.log
done
for wl in c1_fixture_7562 c2_cylinder_100k; do
  for v in "team_pass=0" "team_pass=1"; do
    timeout 300 python scripts/run_probe.py $wl $v > $O/run_${wl}_$v.log 2>&1; echo "$wl $v"; tail -4 $O/run_${wl}_$v.log
  done
done
for v in 1 0 1 0; do
  timeout 300 python bench.py --workload c5_montecarlo_5000 --steps 1 --warmup 1 --repeats 5 --no-configs --no-cpu-baseline --concurrent-pairs 0 --opt pose_copy_stream=$v > $O/c5_$v.json 2>$O/c5_$v.err
  python -c "import json,sys; j=json.load(open('$O/c5_$v.json')); print('c5 pose_copy_stream=$v', round(j['value']), 'it/s', j['ms_per_step'])"
done
timeout 300 python scripts/run_probe.py c4_corridor_1m > $O/run_c4.log 2>&1; tail -4 $O/run_c4.log
