#!/bin/bash
# GPU box: rocprofv3 --pmc passes (one per counter group) of scripts/run_probe.py; prints the counters of the k_lin launches of the
# LAST run by iteration.  usage: pmc_by_iter.sh <workload> "<C1 C2 ..>" ["<C3 C4 ..>" ...]
WL=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_iter; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
g=0
for grp in "$@"; do
  g=$((g+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/g$g -- python $R/scripts/run_probe.py $WL > $O/log_g$g.txt 2>&1
done
python - "$O" <<'PY'
import csv, glob, sys, collections
iters = [0, 1, 2, 3, 4, 5, 8, 10, 12, 14, 16, 18, 20, 22, 25, 30, 40, 49]
for d in sorted(glob.glob(sys.argv[1] + "/g*")):
    per = collections.defaultdict(dict)       # dispatch id -> counter -> value
    for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if "k_lin" in r["Kernel_Name"]:
                per[int(r["Dispatch_Id"])][r["Counter_Name"]] = per[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    ids = sorted(per)
    if len(ids) < 50:
        print(d, "only", len(ids), "k_lin dispatches"); continue
    last = ids[-50:]
    names = sorted({k for i in last for k in per[i]})
    print("%-5s " % "iter" + " ".join("%22s" % n[:22] for n in names))
    for it in iters:
        print("%-5d " % it + " ".join("%22.4g" % per[last[it]].get(n, float("nan")) for n in names))
PY
