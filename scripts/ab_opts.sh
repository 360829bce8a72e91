#!/bin/bash
# A/B of backend options on ONE box, alternating: bench.py restricted to one workload + regimes + converged / cold run.
# usage: scripts/ab_opts.sh <tag> <rounds> <workload> "<opts of variant 0>" "<opts of variant 1>" ...      (opts: "k=v k=v", "" = defaults)
cd "$(dirname "$0")/.."
TAG=$1; ROUNDS=$2; WL=$3; shift 3
O=$PWD/gpurun_out/$TAG; mkdir -p $O
for r in $(seq 1 $ROUNDS); do
  i=0
  for V in "$@"; do
    OPTS=""; for kv in $V; do OPTS="$OPTS --opt $kv"; done
    python bench.py --steps 50 --warmup 50 --repeats 30 --min-seconds 0 --no-cpu-baseline --no-configs --concurrent-pairs 0 --workload $WL $OPTS > $O/v${i}_$r.json 2> $O/v${i}_$r.err
    i=$((i+1))
  done
done
python - "$@" <<PY
import json, glob, os, sys
names = sys.argv[1:]
for f in sorted(glob.glob("$O/v*.json")):
    b = os.path.basename(f)[:-5]
    try:
        d = json.loads(open(f).read().strip().split("\n")[-1])
    except Exception as e:
        print(b, "unreadable", e, open(f[:-5] + ".err").read()[-400:]); continue
    rg = d.get("roofline_by_regime", {})
    bi = rg.get("by_iteration_us", {})
    v = int(b[1:].split("_")[0])
    print("%-6s [%-40s] value %8.1f conv %7.1f it/s (%.3f ms) cold %.3f ms | all_search %.1f transition %.1f settled %.1f | %s" % (
        b, names[v][:40], d["value"], d["converged_run"]["iterations_per_s"], d["converged_run"]["ms_per_run"], d["cold_run"]["ms_per_run"],
        rg.get("all_search", {}).get("mean_us", 0), rg.get("transition", {}).get("mean_us", 0), rg.get("settled", {}).get("mean_us", 0),
        {k[5:]: round(v_) for k, v_ in bi.items()}))
PY
