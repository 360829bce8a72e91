#!/bin/bash
# GPU box: rocprofv3 kernel trace + one PMC pass of P independent pairs in flight on the one GPU (scripts/concurrent_pairs.py).
# usage: profile_concurrent.sh <workload> <P>
WL=${1:-c2_cylinder_100k}; P=${2:-4}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/prof_conc_${WL}_$P; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/scripts/concurrent_pairs.py $WL 1 $P > $O/plain.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/scripts/concurrent_pairs.py $WL $P > $O/trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc -- python $R/scripts/concurrent_pairs.py $WL $P > $O/pmc.log 2>&1
grep -v amdgpu $O/plain.txt | tail -4
python - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
for fn in glob.glob(O + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "k_lin" in r["Name"]:
            print("trace:", r["Name"][:40], "calls", r["Calls"], "avg ns", r["AverageNs"], "min", r["MinNs"], "max", r["MaxNs"])
acc = collections.defaultdict(list)
for fn in glob.glob(O + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "k_lin" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("pmc: %-20s mean per launch %.4g (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
