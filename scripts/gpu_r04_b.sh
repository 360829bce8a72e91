#!/bin/bash
# round 4, GPU call B: new tests, the wave-phase probe, lean kernel A/B at 5 / 6 / 8 waves per SIMD
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04b; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -5 $O/pytest.log
timeout 300 python scripts/wave_phases.py c4_corridor_1m lean_kernel=0 > $O/phases_c4.log 2>&1; tail -3 $O/phases_c4.log
for v in "" occ6 occ8; do
  L=$R/dcreg_amd/lib/libdcreg_hip${v:+_$v}.so
  DCREG_LIB=$L timeout 200 python scripts/run_probe.py c4_corridor_1m > $O/probe_c4_lean_${v:-occ5}.log 2>&1; tail -2 $O/probe_c4_lean_${v:-occ5}.log
done
timeout 200 python scripts/run_probe.py c4_corridor_1m lean_kernel=0 > $O/probe_c4_nolean.log 2>&1; tail -2 $O/probe_c4_nolean.log
timeout 200 python scripts/run_probe.py c2_cylinder_100k > $O/probe_c2.log 2>&1; tail -2 $O/probe_c2.log
timeout 200 python scripts/run_probe.py c3_pk01_200k > $O/probe_c3.log 2>&1; tail -2 $O/probe_c3.log
timeout 300 python scripts/wave_phases.py c4_corridor_1m > $O/phases_c4_lean.log 2>&1; tail -3 $O/phases_c4_lean.log
