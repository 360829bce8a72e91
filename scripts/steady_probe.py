"""Converged-pose linearisations of a bench workload under a profiler (GPU box): 50 ICP iterations to converge, then N more
linearisations along the same settled trajectory.  `rocprofv3 --kernel-trace --stats -- python scripts/steady_probe.py <workload> N`
then shows what each kernel of a certifying launch costs when (almost) every certificate holds."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dcreg_amd
from dcreg_amd import api
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else "c4_corridor_1m"
n_more = int(sys.argv[2]) if len(sys.argv) > 2 else 100
W = bench.WORKLOADS[wl]; scene, n_pts, radius, run_len = W["scene"], W["n"], W["radius"], W["run_len"]
tgt, src = bench.make_pair(scene, n_pts, seed=100)
ctx = dcreg_amd.Context(0)
for kv in sys.argv[3:]:
    k, v = kv.split("="); ctx.set_option(k, float(v))
ctx.set_target(tgt, radius); ctx.set_source(src)
cfg = api.default_config(search_radius=radius, max_iterations=run_len, KAPPA_TARGET=10.0, STD_REG_GAMMA=100.0,
                         CONVERGENCE_THRESH_ROT=0.0, CONVERGENCE_THRESH_TRANS=0.0, use_weight_derivative=W["wd"], always_compute_schur=1)
T_init = bench.initial_pose(scene)
res, logs = ctx.icp_run(T_init, "Ours", cfg)
T = np.eye(4); T[:3, :3] = np.array(res.R[:]).reshape(3, 3); T[:3, 3] = res.t[:]
cfg.max_iterations = n_more
ctx.set_option("time_kernels", 1); ctx.set_option("count_searches", 1)
ctx.kernel_time(reset=True); ctx.launch_stats(reset=True)
t0 = time.perf_counter()
res, logs = ctx.icp_run(T, "Ours", cfg)
el = time.perf_counter() - t0
km, kn = ctx.kernel_time()
st = ctx.launch_stats()
print("%s: %d settled iterations: %.1f us per iteration wall, kernels %.1f us (HIP events around each linearisation), %s" % (
    wl, res.iterations, 1e6 * el / max(res.iterations, 1), 1e3 * km / max(kn, 1), st))
