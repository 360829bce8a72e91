"""gpurun_out/fetch_calib (scripts/microbench/fetch_calib.sh) -> profiles/<tag>_fetch_calibration.{md,json}: measured FETCH_SIZE /
WRITE_SIZE of kernels that move known bytes, per access pattern of k_lin; the factors scripts/summarize_profiles.py applies."""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
src = os.path.join(ROOT, "gpurun_out", "fetch_calib")
known = json.load(open(os.path.join(src, "known.json")))["kernels"]


def counters(sub):
    g = sorted(glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    res = collections.defaultdict(lambda: collections.defaultdict(list))
    if g:
        for r in csv.DictReader(open(g[-1])):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            res[(name, int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return res


fetch, write, rd, wr = counters("fetch"), counters("write"), counters("rdreq"), counters("wrreq")
out = {"tag": tag, "known_bytes": known, "patterns": {}}
md = ["# %s - FETCH_SIZE / WRITE_SIZE calibrated on k_lin's access patterns" % tag, "",
      "`scripts/microbench/fetch_calib.hip` under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes; counters in KB).  Known bytes = what",
      "the kernel's loads / stores address; ratio = counter x 1024 / known.  The factor scripts/summarize_profiles.py multiplies a k_lin FETCH_SIZE",
      "by is 1 / ratio of the pattern that dominates the launch (settled launches: `settled`).", "",
      "| size | pattern | known MB | counter MB | ratio counter / known | factor |", "|---|---|---|---|---|---|"]
pat = [("calib_rows4", "rows4_read_bytes", "FETCH_SIZE", fetch), ("calib_pts16", "pts16_read_bytes", "FETCH_SIZE", fetch),
       ("calib_settled", "settled_read_bytes", "FETCH_SIZE", fetch), ("calib_gather16", "gather16_read_bytes_lines128", "FETCH_SIZE", fetch),
       ("calib_write4", "write4_bytes", "WRITE_SIZE", write)]
for size_key, kb in known.items():
    gx = kb["grid_x"] * 256
    for kname, bkey, cname, table in pat:
        vals = None
        for (name, grid), cs in table.items():
            if name.endswith(kname) and (grid == gx or grid == kb["grid_x"]) and cname in cs:
                vals = cs[cname]
        if not vals:
            continue
        v = vals[-1] * 1024.0                      # the last repetition (caches warm as in a bench loop)
        ratio = v / kb[bkey]
        out["patterns"].setdefault(size_key, {})[kname] = {"known_bytes": kb[bkey], "counter_bytes": v, "ratio": ratio, "factor": 1.0 / ratio if ratio > 0 else None,
                                                           "all_reps_kb": vals}
        md.append("| %s | %s (%s) | %.1f | %.1f | %.3f | %.3f |" % (size_key, kname, cname, kb[bkey] / 1e6, v / 1e6, ratio, 1.0 / ratio if ratio > 0 else float("nan")))
md += ["", "gather16: known = 128 B (one line) per lane; the minimum useful bytes are 16 B per lane (1/8 of that)."]
for nm, tab in (("TCC_EA0_RDREQ", rd), ("TCC_EA0_WRREQ", wr)):
    rows = []
    for (name, grid), cs in sorted(tab.items()):
        if "calib_" in name:
            rows.append("| %s | %d | %s |" % (name, grid, ", ".join("%s %.4g" % (k, v[-1]) for k, v in sorted(cs.items()))))
    if rows:
        md += ["", "## %s request counters (last repetition)" % nm, "", "| kernel | grid | counters |", "|---|---|---|"] + rows
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
open(os.path.join(ROOT, "profiles", "%s_fetch_calibration.md" % tag), "w").write("\n".join(md) + "\n")
json.dump(out, open(os.path.join(ROOT, "profiles", "%s_fetch_calibration.json" % tag), "w"), indent=1)
print("\n".join(md))
