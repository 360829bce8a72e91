"""GPU: per iteration of a 50-iteration C4 run - the RMS residual the engine hints the backend with, the fraction of points searched,
and what the launch lasts in four-wave blocks ("one_wave" = 0) and in one-wave blocks (= 2).  What the host's rule can go by."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench
from dcreg_amd import api

name = sys.argv[1] if len(sys.argv) > 1 else "c4_corridor_1m"
w = bench.WORKLOADS[name]
tgt, src = bench.make_pair(w["scene"], w["n"], 100)
T0 = bench.initial_pose(w["scene"])
cfg = api.default_config(search_radius=w["radius"], max_iterations=w["run_len"], KAPPA_TARGET=10.0, STD_REG_GAMMA=100.0, CONVERGENCE_THRESH_ROT=0.0,
                         CONVERGENCE_THRESH_TRANS=0.0, use_weight_derivative=w["wd"], always_compute_schur=1)
rows = {}
for ow in (0, 2):
    c = api.Context(0)
    c.set_option("one_wave", ow)
    c.set_target(tgt, w["radius"]); c.set_source(src)
    h = c.index_info().cell
    c.set_option("time_kernels", 1); c.set_option("record_launches", 1)
    acc = []
    for rep in range(5):
        c.launch_series(reset=True)
        res, lg = c.icp_run(T0, "Ours", cfg)
        ser = c.launch_series(reset=True)
        if rep > 0:
            acc.append((1e3 * ser["ms"][:res.iterations], ser["searched"][:res.iterations] / ser["points"][:res.iterations], np.array([L.rmse for L in lg[:res.iterations]])))
    rows[ow] = (np.mean([a[0] for a in acc], 0), acc[-1][1], acc[-1][2])
    c.close()
print("cell h = %.4f m" % h)
print("iter   rmse[m]  rmse/h  searched   4-wave us  1-wave us   ratio")
for it in range(len(rows[0][0])):
    print("%4d  %8.4f  %6.3f  %8.4f  %9.1f  %9.1f  %6.3f" % (it, rows[0][2][it], rows[0][2][it] / h, rows[0][1][it], rows[0][0][it], rows[2][0][it], rows[2][0][it] / rows[0][0][it]))
