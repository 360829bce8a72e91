"""Short view of bench.py JSON lines.  usage: print_bench.py file.json [...]"""
import json, sys
for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().split("\n")[-1])
    print(f, "value %.1f" % d["value"], "ms/step mean %.5f median %.5f min %.5f max %.5f" % (d["ms_per_step"], d["ms_per_step_median"], d["ms_per_step_min"], d["ms_per_step_max"]))
    r, v = d["roofline"], d.get("roofline_valu_issue") or {}
    print("  roofline: %.1f GB/s frac %.4f traffic %.3e kernel_us_avg %.2f | valu frac %s floor %s" % (r["achieved"], r["frac"], r["traffic"] or 0, r["kernel_us_avg"], v.get("frac"), v.get("floor_us")))
    print("  concurrent", d.get("concurrent_pairs", {}).get("value"), "cpu", d.get("cpu_baseline", {}).get("value"))
    for k in ("all_search", "transition", "settled"):
        g = (d.get("roofline_by_regime") or {}).get(k) or {}
        if g.get("launches"):
            print("  regime %-10s launches %3d mean %.1f us frac %.3f searched %.4f refitted %.4f" % (k, g["launches"], g["mean_us"], g["frac"], g["mean_searched_frac"], g["mean_refitted_frac"]))
    if d.get("converged_run"):
        c = d["converged_run"]
        print("  converged run: %.1f iterations, %.3f ms per run, %.0f it/s" % (c["iterations_to_convergence"], c["ms_per_run"], c["iterations_per_s"]))
    if d.get("cold_run"):
        c = d["cold_run"]
        print("  cold run:      %.1f iterations, %.3f ms per run, %.0f it/s" % (c["iterations_to_convergence"], c["ms_per_run"], c["iterations_per_s"]))
    if (d.get("roofline_by_regime") or {}).get("by_iteration_us"):
        print("  by iteration (us):", {k[5:]: round(v, 1) for k, v in d["roofline_by_regime"]["by_iteration_us"].items()})
    print("  timed %.1f s in %d repeats" % (d.get("timed_seconds", 0.0), d.get("repeats", 0)))
    for k, c in (d.get("configs") or {}).items():
        if "roofline" in c:
            print("   %-24s %.1f it/s, %.5f ms/step, kernel %.2f us, valu frac %s" % (k, c["value"], c["ms_per_step"], c["roofline"]["kernel_us_avg"], (c.get("roofline_valu_issue") or {}).get("frac")))
            for rk in ("all_search", "transition", "settled"):
                g = (c.get("roofline_by_regime") or {}).get(rk) or {}
                if g.get("launches"):
                    print("      regime %-10s launches %3d mean %.1f us searched %.4f" % (rk, g["launches"], g["mean_us"], g["mean_searched_frac"]))
            for rk in ("converged_run", "cold_run"):
                if c.get(rk):
                    print("      %-13s %.1f iterations, %.3f ms per run, %.0f it/s" % (rk, c[rk]["iterations_to_convergence"], c[rk]["ms_per_run"], c[rk]["iterations_per_s"]))
            if c.get("by_host_threads"):
                print("      by host threads:", {kk: round(vv["value"]) for kk, vv in c["by_host_threads"].items()})
        else:
            print("   %-24s %s" % (k, {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in c.items() if kk in ("ms_total", "ms_set_source", "ms_iterations", "iterations", "cpu_oracle")}))
