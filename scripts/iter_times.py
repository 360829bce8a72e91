"""Per-iteration wall time of one ICP run on a bench workload (GPU box): shows what the first, badly aligned
iterations cost relative to the converged ones."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dcreg_amd import scenes as h
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
for rep in range(3):
    res, logs = ctx.icp_run(T_init, "Ours", cfg)
t = np.array([L.iter_time_ms for L in logs]) * 1e3
print(wl, "iterations", len(t), "per-iteration us:", " ".join("%.0f" % x for x in t))
print("sum %.0f us, mean %.1f, first5 mean %.1f, last20 mean %.1f" % (t.sum(), t.mean(), t[:5].mean(), t[-20:].mean()))
print("trans err per iter:", " ".join("%.3f" % L.trans_error_vs_gt for L in logs[:12]))
