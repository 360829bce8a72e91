#!/bin/bash
# round 5, GPU call E: team pass v2 - tests, registration probe, kernel trace of it
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05e; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_round5.py -x -q > $O/tests_round5.log 2>&1; echo "round5 tests rc $?"; tail -5 $O/tests_round5.log
for v in "team_pass=0" "team_pass=1"; do
  timeout 300 python scripts/reg_probe.py $v > $O/reg_$v.log 2>&1; echo "reg $v rc $?"; cat $O/reg_$v.log
done
for wl in c1_fixture_7562; do
  for v in "team_pass=0" "team_pass=1"; do
    timeout 300 python scripts/run_probe.py $wl $v > $O/run_${wl}_$v.log 2>&1; echo "$wl $v"; tail -4 $O/run_${wl}_$v.log
  done
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_reg -- python $R/scripts/reg_probe.py team_pass=1 > $O/trace_reg.log 2>&1
python $R/scripts/trace_summary.py $O/trace_reg | head -5
