#!/bin/bash
# A/B on the GPU box: product library vs a tagged variant (dcreg_amd/lib/libdcreg_hip_<tag>.so), per-iteration times.
cd "$(dirname "$0")/.."
TAG=$1; shift
O=gpurun_out/r02_ab_$TAG.txt; : > $O
V=$PWD/dcreg_amd/lib/libdcreg_hip_$TAG.so
run() { echo "== $1 | ${2:-product} | $3" >> $O; DCREG_LIB=$2 python scripts/iter_times.py $1 $3 2>&1 | grep -v amdgpu.ids | cut -c1-330 >> $O; }
for wl in c4_corridor_1m c2_cylinder_100k c3_pk01_200k c1_fixture_7562; do
  run $wl "" "$*"; run $wl "$V" "$*"; run $wl "" "$*"; run $wl "$V" "$*"
done
cat $O
