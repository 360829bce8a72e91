#!/bin/bash
# round 5, GPU call A: the advance pass - parity tests first, then per-iteration times of C4 with the pass off / by the rule / forced
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05a; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_round5.py -x -q > $O/tests_round5.log 2>&1; echo "round5 tests rc $?"; tail -5 $O/tests_round5.log
for v in "advance=0" "advance=1" "advance=2"; do
  timeout 300 python scripts/run_probe.py c4_corridor_1m $v > $O/run_c4_$v.log 2>&1; echo "c4 $v rc $?"; tail -2 $O/run_c4_$v.log
done
timeout 300 python scripts/run_probe.py c4_corridor_1m advance=1 advance_lo=0.003 advance_hi=0.7 > $O/run_c4_wide.log 2>&1; tail -2 $O/run_c4_wide.log
for v in "advance=0" "advance=2"; do
  timeout 300 python scripts/run_probe.py c2_cylinder_100k $v > $O/run_c2_$v.log 2>&1; echo "c2 $v"; tail -2 $O/run_c2_$v.log
  timeout 300 python scripts/run_probe.py c3_pk01_200k $v > $O/run_c3_$v.log 2>&1; echo "c3 $v"; tail -2 $O/run_c3_$v.log
done
