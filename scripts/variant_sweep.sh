#!/bin/bash
# A/B of prebuilt library variants (dcreg_amd/lib/variants/lib*.so) on bench workloads (GPU box)
# usage: variant_sweep.sh "wl1 wl2 ..." [bench args]
cd "$(dirname "$0")/.."
WLS=${1:-"c2_cylinder_100k c4_corridor_1m c1_fixture_7562"}; shift
cp dcreg_amd/lib/libdcreg_hip.so /tmp/lib_orig.so
for rep in 1 2; do
for v in dcreg_amd/lib/variants/lib*.so; do
  cp $v dcreg_amd/lib/libdcreg_hip.so
  for wl in $WLS; do
    steps=400; [ $wl = c4_corridor_1m ] && steps=100
    python bench.py --workload $wl --steps $steps --warmup 40 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('$(basename $v)', '$wl', 'it/s %.0f'%d['value'], 'kernel_us %.1f'%d['roofline']['kernel_us_avg'], 'ms/step %.4f'%d['ms_per_step'], 'corr', d['final_stats']['mean_correspondences'], 'terr %.3e'%d['final_stats']['mean_trans_error_m'])"
  done
done
done
cp /tmp/lib_orig.so dcreg_amd/lib/libdcreg_hip.so
