#!/bin/bash
# A/B of two trees on ONE box: this tree against a checkout under _ab/<name> (git worktree of an earlier commit, library built there),
# alternating, bench.py restricted to the headline workload + regimes.  usage: scripts/ab_trees.sh <tag> <name> [rounds] [bench args...]
cd "$(dirname "$0")/.."
TAG=$1; NAME=$2; ROUNDS=${3:-2}; shift 3
O=$PWD/gpurun_out/$TAG; mkdir -p $O
ARGS="--steps 50 --warmup 50 --repeats 40 --no-cpu-baseline --no-configs --concurrent-pairs 0 $*"
for r in $(seq 1 $ROUNDS); do
  (cd _ab/$NAME && python bench.py $ARGS $( [ "$NAME" = r05 ] || echo --min-seconds 0 ) > $O/${NAME}_$r.json 2> $O/${NAME}_$r.err)
  python bench.py $ARGS --min-seconds 0 > $O/new_$r.json 2> $O/new_$r.err
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/*.json")):
    try:
        d = json.loads(open(f).read().strip().split("\n")[-1])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    rg = d.get("roofline_by_regime", {})
    bi = rg.get("by_iteration_us", {})
    print("%-12s value %8.1f conv %7.1f it/s (%.3f ms) cold %s | all_search %.1f transition %.1f settled %.1f | iters %s" % (
        os.path.basename(f)[:-5], d["value"], d["converged_run"]["iterations_per_s"], d["converged_run"]["ms_per_run"],
        ("%.3f ms" % d["cold_run"]["ms_per_run"]) if "cold_run" in d else "-",
        rg.get("all_search", {}).get("mean_us", 0), rg.get("transition", {}).get("mean_us", 0), rg.get("settled", {}).get("mean_us", 0),
        {k[5:]: round(v) for k, v in bi.items()}))
PY
