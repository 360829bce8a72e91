#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05k; mkdir -p $O
cd $R
bash scripts/ab_multi.sh "pfh product" 3 "c4_corridor_1m c3_pk01_200k" > $O/ab.log 2>&1
grep "sum" $O/ab.log | sed 's/.*\(c[0-9]_[a-z0-9_]* [a-z0-9]*\):.*per-iteration us: \([0-9]* [0-9]* [0-9]* [0-9]* [0-9]*\) .*sum \([0-9]*\) us.*/\1 first: \2 sum \3/'
