#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05o; mkdir -p $O
cd $R
bash scripts/ab_multi.sh "r04 product" 3 "c1_fixture_7562 c2_cylinder_100k c3_pk01_200k c4_corridor_1m" > $O/ab.log 2>&1
grep "sum" $O/ab.log | sed 's/.*\(c[0-9]_[a-z0-9_]* [a-z0-9]*\):.*sum \([0-9]*\) us.*/\1 sum \2/'
