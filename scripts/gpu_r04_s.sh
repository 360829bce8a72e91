#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04s; mkdir -p $O
cd $R
timeout 900 python scripts/option_sweep.py c4_corridor_1m 2 '' 'curve_x_scale=0.25' 'curve_x_scale=0.0625' 'curve_x_scale=0.015625' 'curve_x_scale=0.00390625' 2>&1 | grep -v Warn | tee $O/sweep_c4.log
timeout 600 python scripts/option_sweep.py c2_cylinder_100k 2 '' 'curve_x_scale=0.25' 'curve_x_scale=0.0625' 'curve_x_scale=0.015625' 2>&1 | grep -v Warn | tee $O/sweep_c2.log
timeout 600 python scripts/option_sweep.py c3_pk01_200k 2 '' 'curve_x_scale=0.25' 'curve_x_scale=0.0625' 'curve_x_scale=0.015625' 2>&1 | grep -v Warn | tee $O/sweep_c3.log
timeout 600 python scripts/option_sweep.py c1_fixture_7562 2 '' 'curve_x_scale=0.25' 'curve_x_scale=0.0625' 2>&1 | grep -v Warn | tee $O/sweep_c1.log
