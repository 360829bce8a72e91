#!/bin/bash
# A/B of two builds of the library on ONE box: scripts/ab_probe.sh <tagA> <tagB> [rounds] - run_probe on C4 / C2 / C3 / C1 alternately
# (tag "" = the product library), the second run's per-iteration sum of each.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ab; mkdir -p $O
cd $R
A=$1; B=$2; N=${3:-3}
lib() { if [ "$1" = "product" ]; then echo $R/dcreg_amd/lib/libdcreg_hip.so; else echo $R/dcreg_amd/lib/libdcreg_hip_$1.so; fi; }
for wl in c4_corridor_1m c2_cylinder_100k c3_pk01_200k c1_fixture_7562; do
  for r in $(seq 1 $N); do
    for t in $A $B; do
      DCREG_LIB=$(lib $t) timeout 200 python scripts/run_probe.py $wl 2>/dev/null | tail -1 | sed "s/^/$wl $t: /"
    done
  done
done | tee $O/ab_${A}_${B}.log
