#!/bin/bash
# warm-start / cell-size ablation (GPU box): bench lines for the C2 / C4 / C1 workloads
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for wl in c2_cylinder_100k c4_corridor_1m c1_fixture_7562; do
  steps=400; [ $wl = c4_corridor_1m ] && steps=100
  for opts in "warm_start=0 cell_factor=2.0" "warm_start=1 cell_factor=2.0" "warm_start=1 cell_factor=1.5" "warm_start=1 cell_factor=1.2" "warm_start=1 cell_factor=1.0" "warm_start=1 cell_factor=0.8"; do
    o=""; for kv in $opts; do o="$o --opt $kv"; done
    python bench.py --workload $wl --steps $steps --warmup 40 --no-cpu-baseline $o 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('$wl', '$opts', 'it/s %.0f'%d['value'], 'kernel_us %.1f'%d['roofline']['kernel_us_avg'], 'ms/step %.4f'%d['ms_per_step'], 'corr', d['final_stats']['mean_correspondences'], 'terr %.3e'%d['final_stats']['mean_trans_error_m'])"
  done
done
