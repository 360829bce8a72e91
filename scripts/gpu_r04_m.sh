#!/bin/bash
# round 4, GPU call M: when is a start bound worth a probe (far_loose), C4 first iterations
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04m; mkdir -p $O
cd $R
for fl in 1.5 1.0 0.75 2.5; do
  timeout 200 python scripts/run_probe.py c4_corridor_1m far_loose=$fl > $O/probe_c4_fl$fl.log 2>&1; echo "far_loose $fl"; tail -2 $O/probe_c4_fl$fl.log
done
for fl in 1.5 1.0; do
  timeout 200 python scripts/run_probe.py c3_pk01_200k far_loose=$fl > $O/probe_c3_fl$fl.log 2>&1; echo "c3 far_loose $fl"; tail -1 $O/probe_c3_fl$fl.log
  timeout 200 python scripts/run_probe.py c2_cylinder_100k far_loose=$fl > $O/probe_c2_fl$fl.log 2>&1; echo "c2 far_loose $fl"; tail -1 $O/probe_c2_fl$fl.log
done
timeout 600 python -m pytest tests/test_euler_engine.py tests/test_gpu_round4.py -m gpu -x -q 2>&1 | tail -3
