"""Experiment (GPU box): the order of the SOURCE points decides which 64 queries share a wave.  The library sorts along a Hilbert curve
(compact patches); compared here, passed with keep_source_order: rows of the target grid's orientation - (z, y) bins of b x the cell,
x ascending inside a bin - which give the lanes of a wave the same position relative to the (y, z) rows the search walks.
usage: source_order_probe.py workload"""
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
T_init = bench.initial_pose(scene)
cfg = api.default_config(search_radius=radius, max_iterations=run_len, KAPPA_TARGET=10.0, STD_REG_GAMMA=100.0,
                         CONVERGENCE_THRESH_ROT=0.0, CONVERGENCE_THRESH_TRANS=0.0, use_weight_derivative=W["wd"], always_compute_schur=1)


def measure(ctx, label):
    for rep in range(2):
        res, logs = ctx.icp_run(T_init, "Ours", cfg)
    t = np.array([L.iter_time_ms for L in logs]) * 1e3
    print("%-44s iterations 0-5: %s | last: %.0f | run %.0f us" % (label, " ".join("%.0f" % x for x in t[:6]), t[-1], t.sum()), flush=True)


def rows_order(p, b, cell, xb=None):
    mn = p.min(0)
    yb = np.floor((p[:, 1] - mn[1]) / (b * cell)).astype(np.int64)
    zb = np.floor((p[:, 2] - mn[2]) / (b * cell)).astype(np.int64)
    if xb is None:
        return np.lexsort((p[:, 0], yb, zb))
    xc = np.floor((p[:, 0] - mn[0]) / (xb * cell)).astype(np.int64)       # x in chunks: (xchunk, z, y, x)
    return np.lexsort((p[:, 0], yb, zb, xc))


ctx = dcreg_amd.Context(0)
ctx.set_target(tgt, radius)
cell = ctx.index_info().cell
ctx.set_source(src)
measure(ctx, "library order (Hilbert)")
ctx.set_option("keep_source_order", 1)
for b in (0.5, 1.0, 2.0):
    o = rows_order(src, b, cell)
    ctx.set_source(np.ascontiguousarray(src[o]))
    measure(ctx, "rows: (z, y) bins of %.1f cells, x ascending" % b)
for b, xb in ((1.0, 64), (1.0, 16), (2.0, 32)):
    o = rows_order(src, b, cell, xb)
    ctx.set_source(np.ascontiguousarray(src[o]))
    measure(ctx, "x chunks of %d cells, then (z, y) bins of %.1f" % (xb, b))
