#!/bin/bash
# round 4, GPU call H: bench at the driver's arguments with / without the dispatch order; kernel trace of a probe run (events vs trace)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04h; mkdir -p $O
cd $R
for opt in "" "--opt dispatch_order=0"; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-regimes --concurrent-pairs 0 $opt > $O/bench.json 2> $O/bench.err
python -c "import json; j=json.loads(open('$O/bench.json').read().strip().split('\n')[-1]); print('steps20 [$opt]', j['value'], j['ms_per_step'], j['ms_per_step_median'], j['slowest_repeats'], j['roofline']['kernel_us_avg'])"
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_probe -- python $R/scripts/run_probe.py c4_corridor_1m time_kernels=1 record_launches=1 > $O/trace_probe.log 2>&1
tail -2 $O/trace_probe.log
python - <<'PY'
import csv, glob, os
O=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/r04h"
f=sorted(glob.glob(O+"/trace_probe/**/*kernel_trace.csv", recursive=True))[-1]
rows=[r for r in csv.DictReader(open(f)) if "k_lin" in r["Kernel_Name"]]
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows]
print("k_lin dispatches", len(d))
# the probe runs two 50-iteration runs: the last 50 dispatches are the second run
print("trace us, second run:", " ".join("%.0f"%x for x in d[-50:]))
PY
