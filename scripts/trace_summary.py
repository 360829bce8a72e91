"""Per-kernel summary of a rocprofv3 --kernel-trace CSV: calls, mean / median / min / max duration (us), optionally only the last
N dispatches of each kernel.  usage: trace_summary.py <dir or csv> [last_n]"""
import csv, glob, os, sys
import numpy as np
path = sys.argv[1]
last_n = int(sys.argv[2]) if len(sys.argv) > 2 else 0
files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
by = {}
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void dcreg::", "").replace("dcreg::", "")
        by.setdefault(name, []).append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("VGPR_Count", ""), r.get("LDS_Block_Size", "")))
print("%-58s %7s %9s %9s %9s %9s  %s" % ("kernel", "calls", "mean_us", "median", "min", "max", "vgpr/lds"))
for name, v in sorted(by.items(), key=lambda kv: -sum(x[1] for x in kv[1])):
    v.sort()
    if last_n: v = v[-last_n:]
    d = np.array([x[1] for x in v])
    print("%-58s %7d %9.2f %9.2f %9.2f %9.2f  %s/%s" % (name[:58], len(d), d.mean(), np.median(d), d.min(), d.max(), v[0][2], v[0][3]))
if len(sys.argv) > 3 and sys.argv[3] == "seq":      # the last N dispatches of the linearisation kernels in time order
    allv = sorted((x[0], name, x[1]) for name, v in by.items() for x in v if name.startswith(("k_rows", "k_full", "k_search_list")))
    line = []
    for _, name, d in allv[-last_n:]:
        tag = {"k_full": "F", "k_search_list": "S"}.get(name.split("<")[0], "R" if name.endswith("false>") else "L")
        if tag in ("F", "R") and line:
            print(" ".join(line)); line = []
        line.append("%s %.1f" % (tag, d))
    print(" ".join(line))
