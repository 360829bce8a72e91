"""GPU box: what a loose start bound costs the far queries.  The state of a converged C4 run, then linearisations at the run's START
pose with certificates off (every point searched): the first is bounded by stale neighbours (the bench's iteration 0), the second by
the neighbours of that very pose (a perfect bound), the third after a 5 cm step."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dcreg_amd
from dcreg_amd import api, scenes as h
import bench
W = bench.WORKLOADS["c4_corridor_1m"]
tgt, src = bench.make_pair(W["scene"], W["n"], seed=100)
ctx = dcreg_amd.Context(0)
for kv in sys.argv[1:]:
    k, v = kv.split("="); ctx.set_option(k, float(v))
ctx.set_target(tgt, W["radius"]); ctx.set_source(src)
cfg = api.default_config(search_radius=W["radius"], max_iterations=50, KAPPA_TARGET=10.0, STD_REG_GAMMA=100.0,
                         CONVERGENCE_THRESH_ROT=0.0, CONVERGENCE_THRESH_TRANS=0.0, use_weight_derivative=1, always_compute_schur=1)
T0 = bench.initial_pose(W["scene"])
res, logs = ctx.icp_run(T0, "Ours", cfg)
ctx.set_option("use_certificates", 0); ctx.set_option("time_kernels", 1)
prm = api.default_lin_params(W["radius"], 1)
T1 = T0 @ h.pose6d_matrix(0.05, 0.0, 0.0, 0.0, 0.0, 0.0)
for name, T in (("stale bound (iteration 0 of a run)", T0), ("perfect bound (same pose again)", T0), ("after a 5 cm step", T1), ("same again", T1)):
    ctx.kernel_time(reset=True)
    ctx.linearize(T[:3, :3], T[:3, 3], prm)
    ms, n = ctx.kernel_time()
    print("%-40s %.1f us" % (name, 1e3 * ms / max(n, 1)))
