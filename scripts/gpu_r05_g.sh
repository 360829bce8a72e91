#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05g; mkdir -p $O
cd $R
for rep in 1 2; do
for t in product tile8 tile16; do
  if [ $t = product ]; then L=$R/dcreg_amd/lib/libdcreg_hip.so; else L=$R/dcreg_amd/lib/libdcreg_hip_$t.so; fi
  echo "== $t"; DCREG_LIB=$L timeout 300 python scripts/reg_probe.py team_pass=1 2>&1 | grep -E "registration|per-iter|settled"
done
done
