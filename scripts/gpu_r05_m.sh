#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05m; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_c4_adv2 -- python $R/scripts/run_probe.py c4_corridor_1m advance=2 > $O/trace.log 2>&1
python - $O/trace_c4_adv2 <<'PY'
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda x: x[0])
adv = [(s, e) for s, e, n in rows if "k_advance<" in n]
lin = [(s, e) for s, e, n in rows if "k_lin<" in n]
print("k_advance dispatches %d, k_lin %d" % (len(adv), len(lin)))
print("k_advance us (last 49):", " ".join("%.0f" % ((e - s) / 1e3) for s, e in adv[-49:]))
print("k_lin     us (last 50):", " ".join("%.0f" % ((e - s) / 1e3) for s, e in lin[-50:]))
# gap between the end of a pass and the start of the k_lin behind it
gaps = []
for s, e in adv[-49:]:
    nxt = min((ls for ls, le in lin if ls >= e), default=None)
    if nxt: gaps.append((nxt - e) / 1e3)
print("gap pass -> k_lin us: mean %.2f min %.2f max %.2f" % (sum(gaps) / len(gaps), min(gaps), max(gaps)))
PY
