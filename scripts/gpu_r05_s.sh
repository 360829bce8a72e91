#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05s; mkdir -p $O
cd $R
bash scripts/ab_multi.sh "at1024d2 product at1536 at2048" 3 "c4_corridor_1m" > $O/ab.log 2>&1
grep "per-iteration" $O/ab.log | sed 's/.*\(c[0-9]_[a-z0-9_]* [a-z0-9]*\):.*per-iteration us: \(.*\) sum \([0-9]*\) us.*/\1 sum \3 | \2/' | awk '{printf "%s %s %s %s |", $1,$2,$3,$4; for(i=17;i<=27;i++) printf " %s",$i; print ""}'
