#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05t; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_parity.py -x -q 2>&1 | tail -3
for rep in 1 2 3; do
for v in 0 1; do
  echo "gate_in_kernel=$v reg: $(timeout 300 python scripts/reg_probe.py gate_in_kernel=$v 2>&1 | grep -E '^registration')"
  echo "gate_in_kernel=$v c1: $(timeout 300 python scripts/run_probe.py c1_fixture_7562 gate_in_kernel=$v 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-220)"
done
done
