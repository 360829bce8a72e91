"""gpurun_out/prof_<tag>_steady_<workload> (scripts/collect_steady.sh) -> profiles/<tag>_steady_<workload>.{md,json}: the settled launches of a
workload under rocprofv3.  usage: summarize_steady.py <dir> <N> <tag> <workload>"""
import csv, glob, json, os, sys
O, N, tag, wl = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
root = os.path.dirname(os.path.dirname(O))
out = {"tag": tag, "workload": wl, "what": "the last %d k_lin dispatches of scripts/steady_probe.py (settled trajectory)" % N}
md = ["# %s - settled launches of %s under rocprofv3" % (tag, wl), "", "`python scripts/steady_probe.py %s %d`: 50 ICP iterations to converge, then %d more along the settled trajectory; the figures below" % (wl, N, N),
      "are means over the LAST %d dispatches of k_lin in each profiler pass (scripts/collect_steady.sh)." % N, "", "Un-profiled: " + open(os.path.join(O, "plain.log")).read().strip().split("\n")[-1], ""]
def last(sub, pat):
    f = sorted(glob.glob(os.path.join(O, sub, "**", pat), recursive=True), key=os.path.getmtime)
    return f[-1] if f else None
kt = last("trace", "*kernel_trace.csv")
if kt:
    rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(kt)) if "k_lin" in r["Kernel_Name"])[-N:]
    d = [x[1] for x in rows]
    out["kernel_trace_ns"] = {"calls": len(d), "avg": sum(d) / len(d), "min": min(d), "max": max(d)}
    md += ["## Kernel trace", "", "k_lin: %d dispatches, avg %.2f us, min %.2f, max %.2f" % (len(d), sum(d) / len(d) / 1e3, min(d) / 1e3, max(d) / 1e3), ""]
pm = {}
for sub in ("fetch", "write", "sq1", "tcc"):
    f = last(sub, "*counter_collection.csv")
    if not f: continue
    by = {}
    for r in csv.DictReader(open(f)):
        if "k_lin" in r["Kernel_Name"]:
            by.setdefault(r["Counter_Name"], []).append((int(r.get("Dispatch_Id", 0)), float(r["Counter_Value"])))
    for k, v in by.items():
        v.sort(); v = [x[1] for x in v[-N:]]
        pm[k] = sum(v) / len(v)
out["pmc_per_dispatch"] = pm
if pm:
    md += ["## PMC counters, mean per dispatch", "", "| counter | value |", "|---|---|"] + ["| %s | %.4g |" % kv for kv in sorted(pm.items())] + [""]
    if "FETCH_SIZE" in pm:
        raw = (pm["FETCH_SIZE"] + pm.get("WRITE_SIZE", 0)) * 1024; corr = (2 * pm["FETCH_SIZE"] + pm.get("WRITE_SIZE", 0)) * 1024      # (factors: profiles/r04_fetch_calibration.md - read x2.000, write x1.000)
        out["traffic"] = {"fetch_kb": pm["FETCH_SIZE"], "write_kb": pm.get("WRITE_SIZE", 0), "bytes_raw": raw, "bytes_fetch_x2": corr}
        md += ["HBM-side traffic per launch: FETCH_SIZE %.1f KB, WRITE_SIZE %.1f KB -> %.2f MB raw, %.2f MB with the calibrated factors (read x2.000, write x1.000: profiles/r04_fetch_calibration.md)." % (pm["FETCH_SIZE"], pm.get("WRITE_SIZE", 0), raw / 1e6, corr / 1e6), ""]
    if "SQ_WAVES" in pm:
        w = pm["SQ_WAVES"]
        md += ["Per wave: VALU %.0f, SALU %.0f, LDS %.0f, VMEM_RD %.0f instructions; SQ_WAIT_ANY / SQ_WAVE_CYCLES = %.2f" % (
            pm.get("SQ_INSTS_VALU", 0) / w, pm.get("SQ_INSTS_SALU", 0) / w, pm.get("SQ_INSTS_LDS", 0) / w, pm.get("SQ_INSTS_VMEM_RD", 0) / w, pm.get("SQ_WAIT_ANY", 0) / max(pm.get("SQ_WAVE_CYCLES", 1), 1)), ""]
        out["pmc"] = {"SQ_INSTS_VALU_per_launch": pm.get("SQ_INSTS_VALU", 0), "SQ_WAVES_per_launch": w}
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
open(os.path.join(root, "profiles", "%s_steady_%s.md" % (tag, wl)), "w").write("\n".join(md) + "\n")
json.dump(out, open(os.path.join(root, "profiles", "%s_steady_%s.json" % (tag, wl)), "w"), indent=1)
print("\n".join(md))
