#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05n; mkdir -p $O
cd $R
bash scripts/ab_multi.sh "ad2 product ad6" 3 "c4_corridor_1m" > $O/ab.log 2>&1
grep "per-iteration" $O/ab.log | sed 's/.*\(c[0-9]_[a-z0-9_]* [a-z0-9]*\):.*per-iteration us: \(.*\) sum \([0-9]*\) us.*/\1 sum \3 | \2/' | awk '{printf "%s %s %s %s |", $1,$2,$3,$4; for(i=18;i<=27;i++) printf " %s",$i; print ""}'
timeout 300 python -m pytest tests/test_gpu_round5.py -x -q 2>&1 | tail -2
