#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05q; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for t in r04 cb40ca63 product; do
  if [ $t = product ]; then L=$R/dcreg_amd/lib/libdcreg_hip.so; else L=$R/dcreg_amd/lib/libdcreg_hip_$t.so; fi
  rm -rf $O/tr_$t
  DCREG_LIB=$L timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/tr_$t -- python $R/scripts/steady_probe.py c4_corridor_1m 100 > $O/tr_$t.log 2>&1
  echo "$t: $(python $R/scripts/trace_summary.py $O/tr_$t 100 | grep 'k_lin<0, true, true' | awk '{print "last 100 k_lin mean", $5, "median", $6, "min", $7}')"
done
done
cd $R
timeout 300 python -m pytest tests/test_gpu_round5.py -x -q 2>&1 | tail -2
bash scripts/ab_multi.sh "r04 product" 3 "c1_fixture_7562 c4_corridor_1m" > $O/ab.log 2>&1
grep "sum" $O/ab.log | sed 's/.*\(c[0-9]_[a-z0-9_]* [a-z0-9]*\):.*sum \([0-9]*\) us.*/\1 sum \2/'
timeout 300 python scripts/reg_probe.py 2>&1 | grep -E "registration"
