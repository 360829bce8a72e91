#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05p; mkdir -p $O
cd $R
bash scripts/ab_multi.sh "r04 c966d07b cb40ca63 c8a65777 ceacaf0a c5d2609f product" 3 "c1_fixture_7562" > $O/ab.log 2>&1
grep "sum" $O/ab.log | sed 's/.*\(c[0-9]_[a-z0-9_]* [a-z0-9]*\):.*sum \([0-9]*\) us.*/\1 sum \2/'
