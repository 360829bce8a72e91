#!/bin/bash
# GPU box: one rocprofv3 --pmc pass of the one-pair bench loop with the counters given on the command line; prints the mean per
# dispatch of k_lin.  usage: pmc_probe.sh <workload> COUNTER [COUNTER ...]
WL=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_probe; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O -- python $R/bench.py --steps 50 --warmup 50 --repeats 2 --no-cpu-baseline --no-configs --concurrent-pairs 0 --workload $WL > $O/log.txt 2>&1
python - "$O" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for fn in f:
    for r in csv.DictReader(open(fn)):
        if "k_lin" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("%-28s mean %.4g  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
