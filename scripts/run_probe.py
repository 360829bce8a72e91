"""Two back-to-back ICP runs of a bench workload through dcreg_icp_run (GPU box), for a profiler: the second run starts from the
first one's converged state, like bench.py's steady loop."""
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
ctx.set_option("record_launches", 1)
for rep in range(2):
    ctx.launch_series(reset=True)
    res, logs = ctx.icp_run(T_init, "Ours", cfg)
ser = ctx.launch_series(reset=True)
print(wl, "searched permille:", " ".join("%d" % round(1e3 * a / max(b, 1)) for a, b in zip(ser["searched"], ser["points"])))
print(wl, "advance pass     :", "".join("A" if a else "." for a in ser["advanced"]))
t = np.array([L.iter_time_ms for L in logs]) * 1e3
print(wl, "per-iteration us:", " ".join("%.0f" % x for x in t))
print("sum %.0f us, mean %.1f" % (t.sum(), t.mean()))
