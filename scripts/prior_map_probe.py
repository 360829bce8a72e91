"""GPU probe (round 6): one registration of an 8 k-point frame against the seeded prior map (scenes.scene_prior_map) for several index
options - what the dense table's size costs at scale (TLB / cache misses of the table accesses against candidates per query).
usage: prior_map_probe.py [map points] ["k=v k=v" ...]   (each argument one option set; "" = defaults)
PRIOR_EXTENT=<m> in the environment: half the side of the map's square (default 350: 50 M points = 100 per square metre of ground)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dcreg_amd
from dcreg_amd import api, scenes as h

n_map = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
sets = sys.argv[2:] or [""]
extent = float(os.environ.get("PRIOR_EXTENT", "350"))
t_gen = time.perf_counter()
tgt, src = h.scene_prior_map(n_map, extent=extent)
print("map: %d points, %.0f m x %.0f m, generated in %.1f s; frame %d points" % (len(tgt), 2 * extent, 2 * extent, time.perf_counter() - t_gen, len(src)), flush=True)
gt, T0 = h.pose6d_matrix(**h.PK01_GT), h.pose6d_matrix(**h.PK01_INIT)
cfg = api.default_config(search_radius=0.5, max_iterations=30, KAPPA_TARGET=10.0, STD_REG_GAMMA=100.0, CONVERGENCE_THRESH_ROT=1e-5,
                         CONVERGENCE_THRESH_TRANS=1e-3, use_weight_derivative=0, always_compute_schur=1, gt_matrix=gt.reshape(16))
for opts in sets:
    ctx = dcreg_amd.Context(0)
    for kv in opts.split():
        k, v = kv.split("=")
        ctx.set_option(k, float(v))
    t0 = time.perf_counter(); ctx.set_target(tgt, 0.5); t_map = time.perf_counter() - t0
    info = ctx.index_info()
    tt, its = [], []
    t_first = None
    for rep in range(23):
        ta = time.perf_counter()
        ctx.set_source(src)
        res, _ = ctx.icp_run(T0, "Ours", cfg, log_capacity=0)
        if rep == 0:
            t_first = time.perf_counter() - ta
        if rep >= 3:
            tt.append(time.perf_counter() - ta); its.append(res.iterations)
    # a pose 40 m on (and back): what a new window costs once the buffers exist
    t_win = []
    for dx in (40.0, 0.0, 40.0, 0.0):
        Tm = T0.copy(); Tm[0, 3] += dx
        ta = time.perf_counter()
        ctx.linearize(Tm[:3, :3], Tm[:3, 3], api.default_lin_params(0.5, 0))
        t_win.append(1e3 * (time.perf_counter() - ta))
    print("    linearisations 40 m on / back / on / back (each builds a window where one is in use): %s ms" % " ".join("%.1f" % v for v in t_win), flush=True)
    roi = ctx.roi_info()
    print("    first registration %.1f ms (incl. the window build where one is built); window: %s" % (1e3 * t_first, roi), flush=True)
    ctx.set_option("record_launches", 1); ctx.set_option("time_kernels", 1); ctx.launch_series(reset=True)
    ctx.set_source(src); ctx.icp_run(T0, "Ours", cfg, log_capacity=0)
    ser = ctx.launch_series(reset=True)
    ctx.set_option("record_launches", 0); ctx.set_option("time_kernels", 0)
    ctx.set_source(src)
    dd = ctx.linearize(T0[:3, :3], T0[:3, 3], api.default_lin_params(0.5, 0), debug=True)
    ne = (dd["stats"] & 0xFFFF).astype(np.int64)
    print("[%-40s] cell %.3f m, %d cells, build %.0f ms | registration %.3f ms (min %.3f), %d iterations | kernels per launch (us): %s | candidates mean %.0f p90 %d" % (
        opts, info.cell, info.n_cells, 1e3 * t_map, 1e3 * np.mean(tt), 1e3 * np.min(tt), int(np.mean(its)),
        " ".join("%.0f" % (1e3 * m) for m in ser["ms"]), ne.mean(), np.percentile(ne, 90)), flush=True)
    ctx.close()
