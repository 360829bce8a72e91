"""One registration of an 8 k-point frame against the resident 200 k-point map, from host buffers (bench.py c3_registration), timed with
different options: python scripts/reg_probe.py [key=value ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dcreg_amd
from dcreg_amd import api, scenes as h
tgt, src = h.scene_parkinglot()
gt, T0 = h.pose6d_matrix(**h.PK01_GT), h.pose6d_matrix(**h.PK01_INIT)
ctx = dcreg_amd.Context(0)
for kv in sys.argv[1:]:
    k, v = kv.split("="); ctx.set_option(k, float(v))
ctx.set_target(tgt, 0.5)
cfg = api.default_config(search_radius=0.5, max_iterations=30, KAPPA_TARGET=10.0, STD_REG_GAMMA=100.0, CONVERGENCE_THRESH_ROT=1e-5,
                         CONVERGENCE_THRESH_TRANS=1e-3, use_weight_derivative=0, always_compute_schur=1, gt_matrix=gt.reshape(16))
ctx.set_option("record_launches", 1)
t_tot, t_src = [], []
for rep in range(23):
    ta = time.perf_counter(); ctx.set_source(src); tb = time.perf_counter()
    res, logs = ctx.icp_run(T0, "Ours", cfg, log_capacity=0 if rep < 22 else None)
    tc = time.perf_counter()
    if rep >= 3 and rep < 22:
        t_tot.append(tc - ta); t_src.append(tb - ta)
    if rep < 22:
        ctx.launch_series(reset=True)
ser = ctx.launch_series(reset=True)
print("registration: total %.1f us (min %.1f), set_source %.1f us, %d iterations" % (1e6 * np.mean(t_tot), 1e6 * np.min(t_tot), 1e6 * np.mean(t_src), res.iterations))
print("per-iteration us:", " ".join("%.0f" % (1e3 * L.iter_time_ms) for L in logs[:res.iterations]))
print("searched:", " ".join("%d" % x for x in ser["searched"]))
print("refitted:", " ".join("%d" % x for x in ser["refitted"]))
print("pass    :", "".join(".AT"[int(a)] for a in ser["advanced"]))
