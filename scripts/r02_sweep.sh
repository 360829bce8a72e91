#!/bin/bash
# Round-2 A/B on the GPU box: per-iteration times of one C4 run, product library vs the round-1 ring walk, kernel options.
cd "$(dirname "$0")/.."
O=gpurun_out/r02_sweep.txt; : > $O
V1=$PWD/dcreg_amd/lib/libdcreg_hip_v1walk.so
run() { echo "== $1 | $2 | $3" >> $O; DCREG_LIB=$2 python scripts/iter_times.py $1 $3 2>&1 | grep -v amdgpu.ids | cut -c1-330 >> $O; }
for opts in "fast_plane_fit=0 xcd_chunk=0" "fast_plane_fit=1 xcd_chunk=0" "fast_plane_fit=1 xcd_chunk=16"; do
  run c4_corridor_1m "" "$opts"
  run c4_corridor_1m "$V1" "$opts"
done
run c2_cylinder_100k "" "fast_plane_fit=1 xcd_chunk=0"
run c2_cylinder_100k "$V1" "fast_plane_fit=0 xcd_chunk=0"
run c2_cylinder_100k "" "fast_plane_fit=1 xcd_chunk=16"
cat $O
