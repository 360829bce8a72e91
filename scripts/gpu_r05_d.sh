#!/bin/bash
# round 5, GPU call D: kernel trace of the registration probe (team pass forced / off) and of C2 with the team pass
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05d; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in "team_pass=2" "team_pass=0"; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_reg_$v -- python $R/scripts/reg_probe.py $v > $O/trace_reg_$v.log 2>&1
  echo "== reg $v"; python $R/scripts/trace_summary.py $O/trace_reg_$v | head -8
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c2 -- python $R/scripts/run_probe.py c2_cylinder_100k team_pass=1 > $O/trace_c2.log 2>&1
echo "== c2"; python $R/scripts/trace_summary.py $O/trace_c2 | head -8
