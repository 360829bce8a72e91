"""GPU: where a group step of the Monte-Carlo experiment goes, by host threads (DCREG_TRIALS_TIMING: wait for results / host steps / wall)."""
import os, sys
import numpy as np
os.environ["DCREG_TRIALS_TIMING"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from dcreg_amd import api, scenes as h
pts = h.cylinder_cloud()
cfg = api.default_config(search_radius=1.0, max_iterations=30, CONVERGENCE_THRESH_TRANS=1e-3, CONVERGENCE_THRESH_ROT=1e-5, KAPPA_TARGET=10.0, STD_REG_GAMMA=100.0,
                         DEGENERACY_THRES_COND=10.0, DEGENERACY_THRES_EIG=120.0, use_weight_derivative=1, always_compute_schur=1)
base = (0.2, 0.8, 0.5, h.deg2rad(0.1), h.deg2rad(0.1), h.deg2rad(2.0))
ctx = api.Context(0)
ctx.set_target(pts, 1.0); ctx.set_source(pts)
for thr in (16, 2, 2, 4, 16):
    api.set_host_threads(thr)
    print("host threads", thr, flush=True)
    rec, st = ctx.montecarlo_job(base, 2024, 5000, 0.5, np.deg2rad(2.0), "Ours", cfg, slots=256)
    sys.stderr.flush()
ctx.close()
