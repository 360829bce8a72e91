"""Option sweep on one box: for every option set a fresh context, two back-to-back runs of the workload, the second run's sum of
per-iteration times.  usage: option_sweep.py workload reps 'k=v,k=v' 'k=v' ...  ('' = defaults)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dcreg_amd
from dcreg_amd import api
import bench
wl = sys.argv[1]; reps = int(sys.argv[2]); sets = sys.argv[3:]
W = bench.WORKLOADS[wl]; scene, n_pts, radius, run_len = W["scene"], W["n"], W["radius"], W["run_len"]
tgt, src = bench.make_pair(scene, n_pts, seed=100)
cfg = api.default_config(search_radius=radius, max_iterations=run_len, KAPPA_TARGET=10.0, STD_REG_GAMMA=100.0,
                         CONVERGENCE_THRESH_ROT=0.0, CONVERGENCE_THRESH_TRANS=0.0, use_weight_derivative=W["wd"], always_compute_schur=1)
T_init = bench.initial_pose(scene)
res = {s: [] for s in sets}
for r in range(reps):
    for s in sets:
        ctx = dcreg_amd.Context(0)
        for kv in [x for x in s.split(",") if x]:
            k, v = kv.split("="); ctx.set_option(k, float(v))
        ctx.set_target(tgt, radius); ctx.set_source(src)
        for rep in range(2):
            out, logs = ctx.icp_run(T_init, "Ours", cfg)
        t = np.array([L.iter_time_ms for L in logs]) * 1e3
        res[s].append(t.sum())
        del ctx
for s in sets:
    print("%-18s %-40s %s  median %.0f" % (wl, s or "(defaults)", " ".join("%.0f" % x for x in res[s]), np.median(res[s])))
