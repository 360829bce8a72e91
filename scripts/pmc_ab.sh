#!/bin/bash
# GPU box: wave-level instruction counters of k_linearize for the product library and a tagged variant on one workload.
# usage: scripts/pmc_ab.sh <tag> [workload]
R=$(cd "$(dirname "$0")/.." && pwd); TAG=$1; WL=${2:-c4_corridor_1m}
cd /tmp && export TMPDIR=/tmp
for lib in "" "$R/dcreg_amd/lib/libdcreg_hip_$TAG.so"; do
  name=${lib:+$TAG}; name=${name:-product}
  O=$R/gpurun_out/pmc_ab_${name}; rm -rf $O; mkdir -p $O
  DCREG_LIB=$lib timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O -- \
     python $R/bench.py --steps 50 --warmup 50 --repeats 2 --no-cpu-baseline --no-configs --concurrent-pairs 0 --workload $WL > $O/log.txt 2>&1
  python - "$O" "$name" <<'PY'
import sys, glob, csv, collections
d, name = sys.argv[1], sys.argv[2]
f = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))[-1]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "k_linearize" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
w = sum(agg["SQ_WAVES"]) / len(agg["SQ_WAVES"])
print(name, "launches", len(agg["SQ_WAVES"]), "waves %.0f" % w, " ".join("%s/wave %.0f" % (k, sum(v) / len(v) / w) for k, v in sorted(agg.items()) if k != "SQ_WAVES"))
PY
done
