#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python -m pytest tests/test_gpu_round5.py -x -q -k "gated_in or registration or invisible" 2>&1 | tail -2
for rep in 1 2 3; do
for t in sd2 product; do
  if [ $t = product ]; then L=$R/dcreg_amd/lib/libdcreg_hip.so; else L=$R/dcreg_amd/lib/libdcreg_hip_$t.so; fi
  echo "$t c1: $(DCREG_LIB=$L timeout 300 python scripts/run_probe.py c1_fixture_7562 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-200)"
  echo "$t reg pass off: $(DCREG_LIB=$L timeout 300 python scripts/reg_probe.py team_pass=0 2>&1 | grep -E '^registration')"
  echo "$t reg: $(DCREG_LIB=$L timeout 300 python scripts/reg_probe.py 2>&1 | grep -E '^registration')"
done
done
