"""Where does a step's wall time go?  (GPU box only)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as h
import dcreg_amd
from dcreg_amd import api

ctx = dcreg_amd.Context(0)
CF = float(os.environ.get('CF', '2.0'))
ctx.set_option('lds_pad', float(os.environ.get('PAD', '0')))
ctx.set_option('cell_factor', CF)
print('cell_factor', CF)
for n in (7562, 100_000, 1_000_000):
    tgt = h.scene_cylinder(n, seed=1, noise=0.01) if n != 1_000_000 else h.scene_corridor(n, seed=1)
    rng = np.random.default_rng(0)
    src = (tgt + rng.normal(0, 0.01, tgt.shape)).astype(np.float32)
    t0 = time.perf_counter(); ctx.set_target(tgt, 1.0); t1 = time.perf_counter(); ctx.set_source(src); t2 = time.perf_counter()
    T0 = h.pose6d_matrix(0.05, -0.08, 0.03, 0.003, -0.002, 0.008 if n != 1_000_000 else 0.0004)
    R = np.ascontiguousarray(T0[:3, :3]).reshape(9); t = T0[:3, 3].copy()
    prm = api.default_lin_params(1.0, 1); out = api.LinOut()
    for timed, tile in ((1, 1), (0, 1)):
        ctx.set_option("time_kernels", timed); ctx.kernel_time(reset=True)
        for _ in range(20): ctx.linearize_raw(R, t, prm, out)
        ctx.kernel_time(reset=True)
        K = 200 if n < 1_000_000 else 40
        a = time.perf_counter()
        for _ in range(K): ctx.linearize_raw(R, t, prm, out)
        b = time.perf_counter()
        ms, cnt = ctx.kernel_time(reset=True)
        info = ctx.index_info()
        print("n=%8d timed=%d tile=%d  wall/call %.1f us  kernel(evt) %.1f us  n_eff %d  cell %.3f cells %d  build tgt %.1f ms src %.1f ms" % (
            n, timed, tile, (b - a) / K * 1e6, (ms / cnt * 1e3) if cnt else -1, out.n_eff, info.cell, info.n_cells, (t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
