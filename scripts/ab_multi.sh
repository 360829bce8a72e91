#!/bin/bash
# several builds of the library on ONE box, alternating: scripts/ab_multi.sh "<tag tag ...>" [rounds] [workloads]; tag "product" = the product library
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ab; mkdir -p $O
cd $R
TAGS=$1; N=${2:-2}; WLS=${3:-"c4_corridor_1m c2_cylinder_100k c3_pk01_200k c1_fixture_7562"}
lib() { if [ "$1" = "product" ]; then echo $R/dcreg_amd/lib/libdcreg_hip.so; else echo $R/dcreg_amd/lib/libdcreg_hip_$1.so; fi; }
for wl in $WLS; do
  for r in $(seq 1 $N); do
    for t in $TAGS; do
      DCREG_LIB=$(lib $t) timeout 200 python scripts/run_probe.py $wl 2>/dev/null | tail -2 | tr '\n' ' ' | sed "s/^/$wl $t: /"; echo
    done
  done
done | tee $O/ab_multi.log
