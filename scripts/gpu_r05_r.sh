#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2 3; do
for t in prev product; do
  if [ $t = product ]; then L=$R/dcreg_amd/lib/libdcreg_hip.so; else L=$R/dcreg_amd/lib/libdcreg_hip_$t.so; fi
  echo "$t pass on : $(DCREG_LIB=$L timeout 300 python scripts/reg_probe.py 2>&1 | grep -E '^registration')"
  echo "$t pass off: $(DCREG_LIB=$L timeout 300 python scripts/reg_probe.py team_pass=0 2>&1 | grep -E '^registration')"
done
done
