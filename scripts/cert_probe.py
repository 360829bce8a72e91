"""Per-iteration picture of one ICP run on a bench workload (GPU box): how many points each linearisation had to search (the others'
certificates held), the kernel's time (HIP events) and the pose step.  Blocking dcreg_linearize calls + the host step through the solver seam, i.e. the engine loop
unrolled in Python (slower per iteration than dcreg_icp_run, same launches)."""
import os, sys, time
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
ctx.set_option("time_kernels", 1); ctx.set_option("count_searches", 1)
prm = api.default_lin_params(radius, W["wd"])
cfg = api.default_config(search_radius=radius, max_iterations=run_len, KAPPA_TARGET=10.0, STD_REG_GAMMA=100.0,
                         CONVERGENCE_THRESH_ROT=0.0, CONVERGENCE_THRESH_TRANS=0.0, use_weight_derivative=W["wd"], always_compute_schur=1)
T_init = bench.initial_pose(scene)
for rep in range(2):
    T = T_init.copy()
    rows = []
    for it in range(run_len):
        ctx.launch_stats(reset=True); ctx.kernel_time(reset=True)
        t0 = time.perf_counter()
        lo = ctx.linearize(T[:3, :3], T[:3, 3], prm)
        wall = time.perf_counter() - t0
        st = ctx.launch_stats(); km, kn = ctx.kernel_time()
        det, hand = api.METHODS["Ours"]
        an = api.analyze_degeneracy(lo["H"], det, hand, cfg)
        dx = api.solve_degenerate_system(lo["H"], lo["g"], hand, cfg, an)
        R2, t2 = api.boxplus(T[:3, :3], T[:3, 3], dx)
        T[:3, :3], T[:3, 3] = R2, t2
        rows.append((it, st["points_searched"], 1e3 * km / max(kn, 1), 1e6 * wall, np.linalg.norm(dx[3:]), np.linalg.norm(dx[:3])))
print(wl, "n_src", len(src), "cell %.4f" % ctx.index_info().cell)
print(" it  searched  kernel_us  wall_us   |dt|      |dw|")
for r in rows:
    print("%3d %9d %10.1f %8.1f  %.2e %.2e" % r)
k = np.array([r[2] for r in rows])
print("kernel time: sum %.0f us, mean %.1f, first5 mean %.1f, last20 mean %.1f" % (k.sum(), k.mean(), k[:5].mean(), k[-20:].mean()))
