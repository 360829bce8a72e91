#!/bin/bash
# Round-2 option sweep on the GPU box: per-iteration times of one C4 run under the kernel options, C2 / C5 spot checks.
cd "$(dirname "$0")/.."
O=gpurun_out/r02_sweep.txt; : > $O
for opts in "fast_plane_fit=0 xcd_chunk=0" "fast_plane_fit=1 xcd_chunk=0" "fast_plane_fit=1 xcd_chunk=4" "fast_plane_fit=1 xcd_chunk=16" "fast_plane_fit=1 xcd_chunk=64"; do
  echo "== c4 $opts" >> $O
  python scripts/iter_times.py c4_corridor_1m $opts 2>&1 | cut -c1-420 >> $O
done
for opts in "fast_plane_fit=0 xcd_chunk=0" "fast_plane_fit=1 xcd_chunk=0" "fast_plane_fit=1 xcd_chunk=16"; do
  echo "== c2 $opts" >> $O
  python scripts/iter_times.py c2_cylinder_100k $opts 2>&1 | cut -c1-300 >> $O
done
echo "== c5" >> $O
DCREG_TRIALS_TIMING=1 python bench.py --workload c5_montecarlo_fixture --steps 30 --warmup 30 --repeats 5 --no-configs --no-cpu-baseline --concurrent-pairs 0 2>&1 | cut -c1-700 >> $O
cat $O
