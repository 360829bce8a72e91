"""Turn gpurun_out/prof_<tag>_<workload>/ (rocprofv3 CSVs) into profiles/<tag>_<workload>.{md,json}."""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, wl = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", "prof_%s_%s" % (tag, wl))

def one(pattern):
    g = sorted(glob.glob(os.path.join(src, pattern), recursive=True), key=os.path.getmtime)
    return g[-1] if g else None    # newest run (gpurun_out accumulates earlier runs)

def short(name):
    return name.split("(")[0].replace("void ", "").strip()[:60]

def library_kernels():
    """the kernel instantiations dcreg_amd/lib/libdcreg_hip.so exports (demangled, argument lists cut) - a profile whose dcreg:: kernels are
    not among them was taken with another build and must not be summarised under this tree's name"""
    import subprocess
    lib = os.path.join(ROOT, "dcreg_amd", "lib", "libdcreg_hip.so")
    raw = subprocess.run(["strings", "-a", lib], capture_output=True, text=True, check=True).stdout.split("\n")
    syms = sorted({s for s in raw if s.startswith("_Z") and "dcreg" in s})
    dem = subprocess.run(["c++filt"], input="\n".join(syms), capture_output=True, text=True, check=True).stdout.split("\n")
    return {short(d) for d in dem if "dcreg::" in d and "__device_stub__" not in d}

def check_kernels(names):
    have = library_kernels()
    bad = sorted({short(n) for n in names if "dcreg::" in n and short(n) not in have})
    if bad:
        sys.exit("summarize_profiles: the trace holds dcreg kernels this tree's library does not export (stale profile?): %s" % ", ".join(bad))

ALL_LIN = "dcreg::k_lin (all instantiations)"
# one linearisation = one k_lin dispatch + the advance pass that may run in front of it (k_advance / k_advance_team) + k_sum_tiles behind a
# one-wave launch: per-linearisation
# figures are the totals over all of those kernels divided by the number of k_lin dispatches
LIN = "one linearisation (k_lin + advance passes)"
def is_lin(name): return "k_lin" in name or "k_advance" in name or "k_sum_tiles" in name      # (k_sum_tiles: behind k_lin's one-wave launches)

out = {"tag": tag, "workload": wl}
md = ["# %s — rocprofv3 summary, workload %s" % (tag, wl), "",
      "Command: `python bench.py --steps 50 --warmup 50 --repeats 4 --min-seconds 0 --no-cpu-baseline --no-configs --no-regimes --concurrent-pairs 0 --workload %s` (the one-pair timed region only) under", 
      "`rocprofv3 --kernel-trace --stats` and separate `--pmc` passes (scripts/collect_profiles.sh).", ""]
md[2] = md[2] % wl
bp = os.path.join(src, "bench_plain.json")
if os.path.exists(bp) and os.path.getsize(bp):
    try:
        b = json.loads(open(bp).read().strip().split("\n")[-1])
        md[2] = md[2].replace("--steps 50 --warmup 50 --repeats 4", "--steps %d --warmup %d --repeats %d" % (b.get("steps", 50), b.get("warmup", 50), b.get("repeats", 4)))
        out["bench_unprofiled"] = {k: b[k] for k in ("value", "ms_per_step", "roofline", "correspondence_queries_per_s") if k in b}
        md += ["Un-profiled bench line: value %.1f it/s, %.3f ms/step, kernel (HIP events) %.2f us, roofline frac %.4f" % (
            b["value"], b["ms_per_step"], b["roofline"]["kernel_us_avg"], b["roofline"]["frac"]), ""]
    except Exception as e:
        md += ["(bench line unreadable: %s)" % e, ""]
kt = one("trace/**/*kernel_trace.csv")
if kt:
    agg = collections.defaultdict(list)
    check_kernels({r["Kernel_Name"] for r in csv.DictReader(open(kt))})
    for r in csv.DictReader(open(kt)):
        agg[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        if "k_lin" in r["Kernel_Name"]:        # the launches of a run use several instantiations (warm-bound form): one row for all
            agg[ALL_LIN].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    md += ["## Kernel trace (ns)", "", "| kernel | calls | avg | min | max | total |", "|---|---|---|---|---|---|"]
    ks = {}
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        md.append("| %s | %d | %.0f | %d | %d | %d |" % (k, len(v), sum(v) / len(v), min(v), max(v), sum(v)))
        ks[k] = {"calls": len(v), "avg_ns": sum(v) / len(v), "min_ns": min(v), "max_ns": max(v)}
    n_lin = len(agg.get(ALL_LIN, []))
    if n_lin:
        tot = sum(sum(v) for k, v in agg.items() if is_lin(k) and k != ALL_LIN)
        n_pass = sum(len(v) for k, v in agg.items() if "k_advance" in k)
        md.append("| %s | %d | %.0f | | | %d |" % (LIN, n_lin, tot / n_lin, tot))
        ks[LIN] = {"calls": n_lin, "avg_ns": tot / n_lin, "passes": n_pass}
        md += ["", "(%d of the %d linearisations ran an advance pass in front of k_lin; the last row charges the passes to them)" % (n_pass, n_lin)]
    out["kernel_trace"] = ks
    md.append("")
# rocprofv3's own --stats table of the same process
st = kt.replace("kernel_trace.csv", "kernel_stats.csv") if kt else None
if st and os.path.exists(st):
    rows = list(csv.reader(open(st)))
    with open(os.path.join(ROOT, "profiles", "%s_%s_kernel_stats.csv" % (tag, wl)), "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_ALL)
        for r in rows:
            w.writerow([r[0][:160]] + r[1:])      # rocprim template names run to kilobytes
def counters(sub):
    f = one(sub + "/**/*counter_collection.csv")
    res = collections.defaultdict(lambda: collections.defaultdict(list))
    if f:
        check_kernels({r["Kernel_Name"] for r in csv.DictReader(open(f))})
        for r in csv.DictReader(open(f)):
            res[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if "k_lin" in r["Kernel_Name"]:
                res[ALL_LIN][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return res
pm = {}
for sub in ("fetch", "write", "sq1", "sq2", "tcc"):
    cs = counters(sub)
    n_lin_c = 0
    for k, v in cs.items():
        if k == ALL_LIN and v:
            n_lin_c = max(len(vals) for vals in v.values())
    for k, v in cs.items():
        if is_lin(k) or "k_finalize" in k:
            for cn, vals in v.items():
                pm.setdefault(k, {})[cn] = sum(vals) / len(vals)
    if n_lin_c:        # per linearisation: the passes' counters charged to the k_lin dispatches
        tot = collections.defaultdict(float)
        for k, v in cs.items():
            if is_lin(k) and k != ALL_LIN:
                for cn, vals in v.items():
                    tot[cn] += sum(vals)
        for cn, t in tot.items():
            pm.setdefault(LIN, {})[cn] = t / n_lin_c
out["pmc_per_dispatch"] = pm
if pm:
    md += ["## PMC counters, mean per dispatch", ""]
    for k, v in pm.items():
        md.append("**%s**" % k)
        md.append("")
        md += ["| counter | value |", "|---|---|"] + ["| %s | %.4g |" % (a, b) for a, b in sorted(v.items())] + [""]
    lin = pm.get(LIN) or pm.get(ALL_LIN) or next((v for k, v in pm.items() if "k_lin" in k), None)     # mean over ALL linearisations of the run
    if lin and "FETCH_SIZE" in lin:
        fetch_kb, write_kb = lin["FETCH_SIZE"], lin.get("WRITE_SIZE", 0.0)
        raw = (fetch_kb + write_kb) * 1024.0
        # the read / write factors measured on this kernel's own access patterns (scripts/microbench/fetch_calib.hip: 4 B-per-lane state
        # rows, 16 B-per-lane points, their settled mix, 16 B gathers, 4 B-per-lane row writes)
        calf = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_fetch_calibration.json")))
        f_read, f_write, cal_src = 2.0, 1.0, "MI355X_MICROARCH.md (16 B/lane streaming reads; other widths uncalibrated)"
        if calf:
            cal = json.load(open(calf[-1]))["patterns"]
            big = cal.get("n_12000000") or next(iter(cal.values()))
            if "calib_settled" in big and "calib_write4" in big:
                f_read, f_write = big["calib_settled"]["factor"], big["calib_write4"]["factor"]
                cal_src = os.path.relpath(calf[-1], ROOT) + " (rows4 %.3f, pts16 %.3f, settled mix %.3f, gather16 %.3f; write4 %.3f)" % (
                    big["calib_rows4"]["factor"], big["calib_pts16"]["factor"], big["calib_settled"]["factor"], big["calib_gather16"]["factor"], big["calib_write4"]["factor"])
        corr = (f_read * fetch_kb + f_write * write_kb) * 1024.0
        out["traffic"] = {"fetch_kb": fetch_kb, "write_kb": write_kb, "bytes_raw": raw, "bytes_fetch_x2": corr, "read_factor": f_read, "write_factor": f_write,
                          "calibration": cal_src,
                          "note": "FETCH_SIZE/WRITE_SIZE are KB at the L2<->fabric boundary (Infinity-Cache hits included).  On gfx950 FETCH_SIZE "
                                  "reports half the bytes read - measured on this kernel's own patterns, see `calibration`; WRITE_SIZE is exact."}
        md += ["## HBM-side traffic per linearisation (k_lin + advance passes)", "",
               "FETCH_SIZE %.1f KB, WRITE_SIZE %.1f KB -> %.2f MB raw, %.2f MB with the calibrated factors (read x%.3f, write x%.3f: %s)." % (
                   fetch_kb, write_kb, raw / 1e6, corr / 1e6, f_read, f_write, cal_src), ""]
    if lin and "SQ_WAVES" in lin:
        w = lin["SQ_WAVES"]
        md += ["## Per-wave instruction mix, all kernels of a linearisation", "",
               "waves %.0f; per wave: VALU %.0f, SALU %.0f, LDS %.0f, VMEM_RD %.0f; SQ_WAIT_ANY / SQ_WAVE_CYCLES = %.2f" % (
                   w, lin.get("SQ_INSTS_VALU", 0) / w, lin.get("SQ_INSTS_SALU", 0) / w, lin.get("SQ_INSTS_LDS", 0) / w,
                   lin.get("SQ_INSTS_VMEM_RD", 0) / w, lin.get("SQ_WAIT_ANY", 0) / max(lin.get("SQ_WAVE_CYCLES", 1), 1)), ""]
        out["per_wave"] = {"valu": lin.get("SQ_INSTS_VALU", 0) / w, "salu": lin.get("SQ_INSTS_SALU", 0) / w}
        # wave-level VALU instructions of one launch -> the VALU-issue floor bench.py prices the kernel against
        out["pmc"] = {"SQ_INSTS_VALU_per_launch": lin.get("SQ_INSTS_VALU", 0), "SQ_WAVES_per_launch": w}
        mixf = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_valu_mix.json")))
        cyc = json.load(open(mixf[-1]))["mean_cycles_per_valu_w8"] if mixf else 2.0
        floor_us = lin.get("SQ_INSTS_VALU", 0) * cyc / 1024 / 2.4e9 * 1e6
        md += ["VALU-issue floor of one launch: %.3g wave instructions x %.2f cycles (saturated issue cost of the kernel's instruction mix, "
               "scripts/microbench + scripts/asm_mix.py) / 1024 SIMDs / 2.4 GHz = %.1f us" % (lin.get("SQ_INSTS_VALU", 0), cyc, floor_us), ""]
        if "SQ_ACTIVE_INST_VALU" in lin and lin.get("SQ_INSTS_VALU", 0) > 0:
            md += ["Counter cross-check: SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = %.2f (the guide counts SQ_ACTIVE_INST_* in quad-cycles: %.1f cycles per VALU "
                   "instruction at this kernel's occupancy - between the microbenchmark's 2.6 cycles at 4 waves per SIMD and 5.4 for a wave alone)" % (
                       lin["SQ_ACTIVE_INST_VALU"] / lin["SQ_INSTS_VALU"], 4.0 * lin["SQ_ACTIVE_INST_VALU"] / lin["SQ_INSTS_VALU"]), ""]
            out["pmc"]["cycles_per_valu_measured"] = 4.0 * lin["SQ_ACTIVE_INST_VALU"] / lin["SQ_INSTS_VALU"]
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
open(os.path.join(ROOT, "profiles", "%s_%s.md" % (tag, wl)), "w").write("\n".join(md) + "\n")
json.dump(out, open(os.path.join(ROOT, "profiles", "%s_%s.json" % (tag, wl)), "w"), indent=1)
print("\n".join(md))
