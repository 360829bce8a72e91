import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
"""GPU: the Monte-Carlo experiment with the host threads bench.py sets, or (--default-threads) with whatever OpenMP defaults to - the
engines clamp their teams to the CPUs the cgroup grants (engine.cpp usable_cpus)."""
from dcreg_amd import api, hostinfo, scenes as h
if "--default-threads" not in sys.argv:
    api.set_host_threads(hostinfo.setup_rank(0, 1)[0])
print("host threads (OpenMP):", api.load().dcreg_get_host_threads(), flush=True)
pts = h.cylinder_cloud()
cfg = api.default_config(search_radius=1.0, max_iterations=30, CONVERGENCE_THRESH_TRANS=1e-3, CONVERGENCE_THRESH_ROT=1e-5, KAPPA_TARGET=10.0, STD_REG_GAMMA=100.0,
                         DEGENERACY_THRES_COND=10.0, DEGENERACY_THRES_EIG=120.0, use_weight_derivative=1, always_compute_schur=1)
base = (0.2, 0.8, 0.5, h.deg2rad(0.1), h.deg2rad(0.1), h.deg2rad(2.0))
ctx = api.Context(0)
ctx.set_target(pts, 1.0); ctx.set_source(pts)
THREADS = [int(x) for x in os.environ.get("MC_THREADS", "0").split(",")]
for thr, slots in [(t, sl) for t in THREADS for sl in (1024, 1536, 2048, 3072, 1024)]:
    if thr: api.set_host_threads(thr)
    ctx.montecarlo_job(base, 2024, 5000, 0.5, np.deg2rad(2.0), "Ours", cfg, slots=slots, want_records=False)
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        _, st = ctx.montecarlo_job(base, 2024, 5000, 0.5, np.deg2rad(2.0), "Ours", cfg, slots=slots, want_records=False)
        ts.append(time.perf_counter() - t0)
    print("threads %2d slots %4d: %.2f ms  %.3f M it/s" % (thr, slots, 1e3 * min(ts), st["iterations_total"] / min(ts) / 1e6), flush=True)
ctx.close()
