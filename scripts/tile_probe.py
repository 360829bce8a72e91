"""GPU probe (round 6): the shared-tile search of dense far waves against the lock-step search, launch by launch of a C4 run.
For the poses of iterations 0..K of the bench's corridor run, from a cold state: candidates per query (lock-step: the lane's own; tile: the
wave's tile), the search phase of every wave in shader clocks (MODE 2 stamps), and the launch time, with tile_search 0 / 1 / 2.
usage: tile_probe.py [workload] [iterations] [extra opts k=v ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from dcreg_amd import api

wl = sys.argv[1] if len(sys.argv) > 1 else "c4_corridor_1m"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
extra = [a.split("=") for a in sys.argv[3:]]
W = bench.WORKLOADS[wl]
tgt, src = bench.make_pair(W["scene"], W["n"], seed=100)
T0 = bench.initial_pose(W["scene"])
prm = api.default_lin_params(W["radius"], W["wd"])
cfg = api.default_config(search_radius=W["radius"], max_iterations=K + 1, KAPPA_TARGET=10.0, STD_REG_GAMMA=100.0, CONVERGENCE_THRESH_ROT=0.0,
                         CONVERGENCE_THRESH_TRANS=0.0, use_weight_derivative=W["wd"], always_compute_schur=1)
ctx = api.Context(0)
ctx.set_target(tgt, W["radius"]); ctx.set_source(src)
res, logs = ctx.icp_run(T0, "Ours", cfg)
poses = [T0] + [np.array(L.transform_matrix[:]).reshape(4, 4) for L in logs[:K]]
ctx.close()
print("workload %s, cell %.4f m" % (wl, 0.0))
for mode, opts in (("off", {"tile_search": 0}), ("rule", {"tile_search": 1}), ("forced", {"tile_search": 2, "tile_max_pts": 1 << 20})):
    c = api.Context(0)
    for k, v in opts.items():
        c.set_option(k, v)
    for k, v in extra:
        c.set_option(k, float(v))
    c.set_option("count_searches", 1)
    c.set_target(tgt, W["radius"]); c.set_source(src)
    for it, T in enumerate(poses):
        R, t = T[:3, :3], T[:3, 3]
        # (a) the warm trajectory as a run has it: one plain launch per pose; timed with events
        c.set_option("time_kernels", 1); c.kernel_time(reset=True); c.launch_stats(reset=True)
        c.linearize(R, t, prm)
        ms, n = c.kernel_time(reset=True); st = c.launch_stats(reset=True)
        # (b) the same pose again from the state (a) left is useless (certificates hold): the stamps and the dump come from a FRESH context state
        c.set_option("time_kernels", 0)
        line = "%-6s it %d: launch %.1f us, searched %d tile %d" % (mode, it, 1e3 * ms / max(n, 1), st["points_searched"], st["points_tile"])
        print(line, flush=True)
    c.close()
# per-wave search phase and candidates, cold state, pose of iteration 0 and 2
for it in (0, 2, 4):
    if it >= len(poses):
        continue
    T = poses[it]; R, t = T[:3, :3], T[:3, 3]
    for mode, opts in (("off", {"tile_search": 0}), ("rule", {"tile_search": 1}), ("forced", {"tile_search": 2, "tile_max_pts": 1 << 20})):
        c = api.Context(0)
        for k, v in opts.items():
            c.set_option(k, v)
        for k, v in extra:
            c.set_option(k, float(v))
        c.set_target(tgt, W["radius"]); c.set_source(src)
        d = c.linearize(R, t, prm, debug=True)
        ne = (d["stats"] & 0xFFFF).astype(np.int64)
        c.reset_warm_state(-1)
        _, stamps = c.linearize_stamped(R, t, prm)
        s = stamps[stamps[:, 2] > 0]
        srch = (s[:, 2] - s[:, 1]).astype(np.int64)
        tot = (s[:, 5] - s[:, 0]).astype(np.int64)
        pc = lambda a: "mean %.0f p50 %d p90 %d p99 %d max %d" % ((a.mean(),) + tuple(np.percentile(a, [50, 90, 99, 100])))
        print("pose %d cold %-6s candidates per point: %s | search phase cycles per wave: %s | whole wave: %s" % (it, mode, pc(ne), pc(srch), pc(tot)), flush=True)
        c.close()
