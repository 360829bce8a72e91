#!/bin/bash
# The scaling curve of BASELINE.json's metric on one node, without touching code: python bench.py --gpus N for N = 1 2 4 8 (as many as the
# node has), one JSON line each under gpurun_out/scale/, then a table - C4 weak scaling (one 1 M x 1 M pair per GPU: `value`) and C5 strong
# scaling (the 5000-trial Monte-Carlo experiment sharded k = rank mod N, records gathered over RCCL: configs.c5_montecarlo_5000) with the
# number of ranks RCCL actually carried (rccl_ranks_seen).  Besides the table: ONE compact JSON line per N on stdout ({"scale_curve": ...}) and
# gpurun_out/scale/curve.jsonl.  bench.py itself exits non-zero when the experiment gathered records from fewer ranks than the job has.
# usage: scripts/scale_curve.sh [steps] [warmup]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/scale; mkdir -p $O
STEPS=${1:-20}; WARMUP=${2:-5}
cd $R
HAVE=$(python -c "import torch; print(torch.cuda.device_count())")
for N in 1 2 4 8; do
  if [ "$N" -gt "$HAVE" ]; then echo "N=$N: only $HAVE device(s) visible - skipped (nothing is reported under that name)"; continue; fi
  HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 1800 python bench.py --gpus $N --steps $STEPS --warmup $WARMUP > $O/n$N.json 2> $O/n$N.err || echo "N=$N failed: $(tail -2 $O/n$N.err)"
done
python - "$O" <<'PY'
import json, os, sys
o = sys.argv[1]
rows = []
for n in (1, 2, 4, 8):
    f = os.path.join(o, "n%d.json" % n)
    if not os.path.exists(f) or not os.path.getsize(f):
        continue
    try:
        d = json.loads(open(f).read().strip().split("\n")[-1])
    except Exception:
        continue
    c5 = (d.get("configs") or {}).get("c5_montecarlo_5000") or {}
    rows.append((n, d["value"], d["ms_per_step"], d["roofline"]["frac"], c5.get("value"), c5.get("ms_per_step"), c5.get("rccl_ranks_seen")))
with open(os.path.join(o, "curve.jsonl"), "w") as out:
    for n, v4, ms4, fr, v5, ms5, seen in rows:
        line = json.dumps({"scale_curve": "dcreg-mi355x", "n_gpus": n, "c4_weak_iterations_per_s": v4, "c4_ms_per_step": ms4, "c4_hbm_frac_rank0": fr,
                           "c5_strong_iterations_per_s": v5, "c5_ms_per_experiment": ms5, "rccl_ranks_seen": seen,
                           "c4_efficiency_vs_1gpu": v4 / (n * rows[0][1] / rows[0][0]), "c5_speedup_vs_first": (v5 / rows[0][4]) if v5 and rows[0][4] else None})
        print(line); out.write(line + "\n")
if rows:
    b4, b5 = rows[0][1], rows[0][4]
    print("| GPUs | C4 weak: it/s (all GPUs) | ms/step | HBM frac (rank 0) | vs N x 1-GPU | C5 strong: it/s | ms / experiment | speed-up | RCCL ranks seen |")
    print("|---|---|---|---|---|---|---|---|---|")
    for n, v4, ms4, fr, v5, ms5, seen in rows:
        print("| %d | %.0f | %.4f | %.3f | %.2f | %s | %s | %s | %s |" % (n, v4, ms4, fr, v4 / (n * b4 / rows[0][0]), "%.0f" % v5 if v5 else "-", "%.2f" % ms5 if ms5 else "-",
                                                                    "%.2f" % (v5 / b5) if v5 and b5 else "-", seen if seen is not None else "-"))
PY
