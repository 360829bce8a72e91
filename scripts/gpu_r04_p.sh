#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04p; mkdir -p $O
cd $R
for n in 2 3 4 6; do
  timeout 300 python bench.py --no-cpu-baseline --no-configs --no-regimes --concurrent-pairs $n > $O/conc$n.json 2> $O/conc$n.err
  python - <<PY
import json
d=json.loads(open("$O/conc$n.json").read().strip().split("\n")[-1])
print("pairs $n: single value %.0f, concurrent %s" % (d["value"], {k: (round(v,1) if isinstance(v,float) else v) for k,v in d.get("concurrent_pairs",{}).items() if k!="note"}))
PY
done
