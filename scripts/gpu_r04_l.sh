#!/bin/bash
# round 4, GPU call L: the gate folded into k_lin; fused batches A/B on the Monte-Carlo experiment
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04l; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -5 $O/pytest.log
for wl in c4_corridor_1m c2_cylinder_100k c3_pk01_200k c1_fixture_7562; do
  timeout 200 python scripts/run_probe.py $wl > $O/probe_$wl.log 2>&1; tail -2 $O/probe_$wl.log
done
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_steps20.json 2> $O/bench_steps20.err; echo "bench rc $?"
python - <<'PY'
import json
j=json.loads(open('/root/repo/gpurun_out/r04l/bench_steps20.json').read().strip().split('\n')[-1])
print("value", j["value"], "frac", j["roofline"]["frac"], "kernel_us", j["roofline"]["kernel_us_avg"], "gate wait", j.get("gate_wait_us_avg"))
r=j.get("roofline_by_regime",{})
for k in ("all_search","transition","settled"):
    print(k, {kk: (round(vv,3) if isinstance(vv,float) else vv) for kk,vv in r.get(k,{}).items()})
print("by_iteration", r.get("by_iteration_us"))
for k,v in j["configs"].items():
    print(k, {kk:vv for kk,vv in v.items() if kk in ("value","ms_per_step","ms_total","ms_set_source","iterations","by_host_threads","ms_iterations")})
PY
