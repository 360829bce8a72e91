#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04o; mkdir -p $O
cd $R
timeout 900 python scripts/option_sweep.py c4_corridor_1m 2 '' 'cell_factor=1.6' 'cell_factor=1.8' 'cell_factor=2.2' 'cell_factor=2.5' 'x_subdiv=4' 'x_subdiv=16' 'xcd_chunk=8' 'xcd_chunk=32' 'cert_inflate=0.002' 'cert_inflate=0.01' 'cert_inflate=0.02' 'team_search=4' 'team_search=12' 'dispatch_order=0' 2>&1 | grep -v Warn | tee $O/sweep_c4.log
timeout 600 python scripts/option_sweep.py c2_cylinder_100k 2 '' 'cell_factor=1.6' 'cell_factor=1.8' 'cell_factor=2.2' 'cell_factor=2.5' 'x_subdiv=4' 'x_subdiv=16' 'cert_inflate=0.002' 'cert_inflate=0.01' 'cert_inflate=0.02' 2>&1 | grep -v Warn | tee $O/sweep_c2.log
timeout 600 python scripts/option_sweep.py c3_pk01_200k 2 '' 'cell_factor=1.6' 'cell_factor=1.8' 'cell_factor=2.2' 'cell_factor=2.5' 'x_subdiv=4' 'x_subdiv=16' 'cert_inflate=0.002' 'cert_inflate=0.01' 'cert_inflate=0.02' 2>&1 | grep -v Warn | tee $O/sweep_c3.log
