#!/bin/bash
# A/B of tagged builds of the library (DCREG_BUILD_TAG / DCREG_EXTRA_FLAGS, dcreg_amd/build.py) on ONE box, alternating: bench.py
# restricted to one workload + regimes + converged / cold run.    usage: scripts/ab_libs.sh <tag> <rounds> <workload> <libtag|product> ...
cd "$(dirname "$0")/.."
TAG=$1; ROUNDS=$2; WL=$3; shift 3
O=$PWD/gpurun_out/$TAG; mkdir -p $O
lib() { if [ "$1" = "product" ]; then echo $PWD/dcreg_amd/lib/libdcreg_hip.so; else echo $PWD/dcreg_amd/lib/libdcreg_hip_$1.so; fi; }
for r in $(seq 1 $ROUNDS); do
  for V in "$@"; do
    DCREG_LIB=$(lib $V) python bench.py --steps 50 --warmup 50 --repeats 30 --min-seconds 0 --no-cpu-baseline --no-configs --concurrent-pairs 0 --workload $WL > $O/${V}_$r.json 2> $O/${V}_$r.err
  done
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/*.json")):
    b = os.path.basename(f)[:-5]
    try:
        d = json.loads(open(f).read().strip().split("\n")[-1])
    except Exception as e:
        print(b, "unreadable", e, open(f[:-5] + ".err").read()[-400:]); continue
    rg = d.get("roofline_by_regime", {})
    bi = rg.get("by_iteration_us", {})
    print("%-12s value %8.1f conv %7.1f it/s (%.3f ms) cold %.3f ms | all_search %.1f transition %.1f settled %.1f | %s" % (
        b, d["value"], d["converged_run"]["iterations_per_s"], d["converged_run"]["ms_per_run"], d["cold_run"]["ms_per_run"],
        rg.get("all_search", {}).get("mean_us", 0), rg.get("transition", {}).get("mean_us", 0), rg.get("settled", {}).get("mean_us", 0),
        {k[5:]: round(v_) for k, v_ in bi.items()}))
PY
