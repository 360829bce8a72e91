#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05f; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_round5.py -x -q > $O/tests_round5.log 2>&1; echo "round5 tests rc $?"; tail -3 $O/tests_round5.log
timeout 300 python scripts/reg_probe.py team_pass=1 > $O/reg.log 2>&1; cat $O/reg.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_reg -- python $R/scripts/reg_probe.py team_pass=1 > $O/trace_reg.log 2>&1
python $R/scripts/trace_summary.py $O/trace_reg | head -5
