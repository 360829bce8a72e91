"""Per-phase shader-clock breakdown of one linearisation on a bench workload (GPU box, instrumented kernel): mean cycles per wave of
prologue / search (A, B, post) / plane fit + row / wave reduction / block reduction.  usage: phase_probe.py [workload] [key=value ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import dcreg_amd
from dcreg_amd import api
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else "c2_cylinder_100k"
W = bench.WORKLOADS[wl]
tgt, src = bench.make_pair(W["scene"], W["n"], seed=100)
ctx = dcreg_amd.Context(0)
for kv in sys.argv[2:]:
    k, v = kv.split("="); ctx.set_option(k, float(v))
ctx.set_target(tgt, W["radius"]); ctx.set_source(src)
cfg = api.default_config(search_radius=W["radius"], max_iterations=W["run_len"], KAPPA_TARGET=10.0, STD_REG_GAMMA=100.0,
                         CONVERGENCE_THRESH_ROT=0.0, CONVERGENCE_THRESH_TRANS=0.0, use_weight_derivative=W["wd"], always_compute_schur=1)
res, logs = ctx.icp_run(bench.initial_pose(W["scene"]), "Ours", cfg)
T = np.array(logs[-1].transform_matrix[:]).reshape(4, 4)
prm = api.default_lin_params(W["radius"], W["wd"])
for rep in range(2):
    out = ctx.linearize(T[:3, :3], T[:3, 3], prm, debug=True)
ck = out["clocks"][: (len(src) + 63) // 64].astype(np.int64)
ph = [ck[:, 1] - ck[:, 0], ck[:, 2] - ck[:, 1], ck[:, 3] - ck[:, 2], ck[:, 4] - ck[:, 3], ck[:, 5] - ck[:, 4]]
pa, pb = ck[:, 6] & 0xFFFFF, (ck[:, 6] >> 20) & 0xFFFFF
print("%s converged pose, instrumented kernel: wave lifetime %d cycles = prologue %d + search %d (tables+list %d, candidate loop %d, rings/re-gather/order %d) "
      "+ plane fit and row %d + wave reduction %d + block reduction and ticket %d" % (
          wl, (ck[:, 5] - ck[:, 0]).mean(), ph[0].mean(), ph[1].mean(), pa.mean(), pb.mean(), (ph[1] - pa - pb).mean(), ph[2].mean(), ph[3].mean(), ph[4].mean()))
