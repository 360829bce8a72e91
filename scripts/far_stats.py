"""Candidates evaluated per query in the first iterations of a bench run (GPU box): the run once (state of the converged pose), then the
poses of its first iterations again through the debug launch (searches every point, bounded by the state; statistics per point)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dcreg_amd
from dcreg_amd import api
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "c4_corridor_1m"
W = bench.WORKLOADS[wl]; scene, n_pts, radius, run_len = W["scene"], W["n"], W["radius"], W["run_len"]
tgt, src = bench.make_pair(scene, n_pts, seed=100)
ctx = dcreg_amd.Context(0)
for kv in sys.argv[2:]:
    k, v = kv.split("="); ctx.set_option(k, float(v))
ctx.set_target(tgt, radius); ctx.set_source(src)
cfg = api.default_config(search_radius=radius, max_iterations=run_len, KAPPA_TARGET=10.0, STD_REG_GAMMA=100.0,
                         CONVERGENCE_THRESH_ROT=0.0, CONVERGENCE_THRESH_TRANS=0.0, use_weight_derivative=W["wd"], always_compute_schur=1)
T_init = bench.initial_pose(scene)
res, logs = ctx.icp_run(T_init, "Ours", cfg)
poses = [T_init] + [np.array(L.transform_matrix[:]).reshape(4, 4) for L in logs[:-1]]
prm = api.default_lin_params(radius, W["wd"])
for k in (0, 1, 2, 3, 4, 5, 8):
    T = poses[k]
    out = ctx.linearize(T[:3, :3], T[:3, 3], prm, debug=True)
    st = out["stats"].astype(np.int64)
    ev, sh = st & 0xFFFF, st >> 16
    # (per point in the caller's order: which 64 share a wave is the library's curve order, not visible here)
    print("iter %d: candidates per query mean %.0f median %d p90 %d p99 %d max %d" % (k, ev.mean(), np.median(ev), np.percentile(ev, 90), np.percentile(ev, 99), ev.max()))
# which queries are the expensive ones (iteration 0): candidates against the distance of their 5th neighbour
T = poses[0]
out = ctx.linearize(T[:3, :3], T[:3, 3], prm, debug=True)
ev = (out["stats"].astype(np.int64) & 0xFFFF)
d5 = np.sqrt(np.maximum(out["nn_d2"].reshape(-1, 5)[:, 4], 0))
d1 = np.sqrt(np.maximum(out["nn_d2"].reshape(-1, 5)[:, 0], 0))
print("iteration 0: 5th-neighbour distance of the queries: median %.3f p90 %.3f p99 %.3f m (cell %.3f)" % (np.median(d5[np.isfinite(d5)]), np.percentile(d5[np.isfinite(d5)], 90), np.percentile(d5[np.isfinite(d5)], 99), ctx.index_info().cell))
edges = [0, 0.05, 0.1, 0.15, 0.2, 0.3, 0.4, 0.6, 0.8, 1.0, 10]
for a, b in zip(edges[:-1], edges[1:]):
    m = (d5 >= a) & (d5 < b)
    if m.any():
        print("  d5 in [%.2f, %.2f): %7d queries, candidates mean %.0f p90 %d, nearest neighbour at %.3f m (mean)" % (a, b, m.sum(), ev[m].mean(), np.percentile(ev[m], 90), d1[m].mean()))
heavy = ev > 200
print("  queries with > 200 candidates: %d (%.2f %%); their d5 mean %.3f, d1 mean %.3f; ideal candidates for a ball of radius d5 on one wall at this density: %.0f" % (
    heavy.sum(), 100.0 * heavy.mean(), d5[heavy].mean(), d1[heavy].mean(), 347 * np.pi * (d5[heavy] ** 2).mean()))
