#!/bin/bash
# round 5, GPU call J: the start-bound probe's scan in batches of 1 (as before) / 2 (product) / 4 trips
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05j; mkdir -p $O
cd $R
bash scripts/ab_multi.sh "pb1 product pb4" 2 "c4_corridor_1m c3_pk01_200k c2_cylinder_100k c1_fixture_7562" > $O/ab.log 2>&1
grep "sum" $O/ab.log | sed 's/.*\(c[0-9]_[a-z0-9_]* [a-z0-9]*\):.*per-iteration us: \([0-9]* [0-9]* [0-9]* [0-9]*\) .*sum \([0-9]*\) us.*/\1 first: \2 sum \3/'
for t in pb1 product pb4; do
  if [ $t = product ]; then L=$R/dcreg_amd/lib/libdcreg_hip.so; else L=$R/dcreg_amd/lib/libdcreg_hip_$t.so; fi
  echo "== reg $t"; DCREG_LIB=$L timeout 300 python scripts/reg_probe.py team_pass=0 2>&1 | grep -E "registration|per-iter"
  DCREG_LIB=$L timeout 300 python scripts/reg_probe.py team_pass=1 2>&1 | grep -E "registration"
done
