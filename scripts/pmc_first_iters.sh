#!/bin/bash
# GPU box: SQ counters of the k_lin dispatches of scripts/run_probe.py (two 50-iteration runs), per dispatch, for the first iterations
# of the second run.  usage: scripts/pmc_first_iters.sh <workload> [option=value ...]
WL=${1:-c4_corridor_1m}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_first_$(echo "$WL $*" | tr ' =' '__'); mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/scripts/run_probe.py $WL $*"
rocprofv3 -L 2>/dev/null | grep -o "SQC\?_[A-Z_]*\(ICACHE\|IFETCH\|WAIT_INST\|INST_LEVEL\)[A-Z_]*" | sort -u > $O/avail.txt
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d $O/p1 -- $CMD > $O/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_IFETCH SQ_INST_CYCLES_VMEM --output-format csv -d $O/p2 -- $CMD > $O/p2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES --output-format csv -d $O/p3 -- $CMD > $O/p3.log 2>&1
python - "$O" <<'PY'
import csv, glob, os, sys
O = sys.argv[1]
print("available:", open(os.path.join(O, "avail.txt")).read().split())
for sub in ("p1", "p2", "p3"):
    f = sorted(glob.glob(os.path.join(O, sub, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    if not f:
        print(sub, "no counters:", open(os.path.join(O, sub + ".log")).read()[-400:]); continue
    by = {}
    for r in csv.DictReader(open(f[-1])):
        if "k_lin" in r["Kernel_Name"]:
            by.setdefault(r["Counter_Name"], {}).setdefault(int(r["Dispatch_Id"]), 0.0)
            by[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    for k, v in sorted(by.items()):
        ids = sorted(v)
        sel = ids[50:56] + ids[-2:] if len(ids) >= 100 else ids[:8]
        print("%-22s" % k, " ".join("%12.4g" % v[i] for i in sel))
PY
