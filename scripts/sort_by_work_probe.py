"""Experiment (GPU box): how much would ordering the queries of a wave by their candidate count buy?
Orders: (a) the library's Hilbert order, (b) a numpy Morton order passed with keep_source_order, (c) Morton chunks of
CH points re-sorted by the candidate count each point had in a previous linearisation at a nearby pose."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dcreg_amd import scenes as h
import dcreg_amd
from dcreg_amd import api
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else "c2_cylinder_100k"
W = bench.WORKLOADS[wl]; scene, n_pts, radius, run_len = W["scene"], W["n"], W["radius"], W["run_len"]
tgt, src = bench.make_pair(scene, n_pts, seed=100)
T0 = h.pose6d_matrix(0.004, -0.003, 0.002, 0.0002, -0.0001, 0.0004)     # a nearly converged pose
T1 = h.pose6d_matrix(0.003, -0.002, 0.001, 0.0001, -0.0001, 0.0003)
prm = api.default_lin_params(radius, 1)


def morton(p):
    q = ((p - p.min(0)) / (p.max(0) - p.min(0) + 1e-9) * 1023).astype(np.uint64)
    def spread(v):
        v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249
        return v
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)


def measure(ctx, label, K=300):
    out = api.LinOut()
    R0, t0 = np.ascontiguousarray(T0[:3, :3]).reshape(9), T0[:3, 3].copy()
    R1, t1 = np.ascontiguousarray(T1[:3, :3]).reshape(9), T1[:3, 3].copy()
    for _ in range(20):
        ctx.linearize_raw(R0, t0, prm, out); ctx.linearize_raw(R1, t1, prm, out)
    a = time.perf_counter()
    for _ in range(K // 2):
        ctx.linearize_raw(R0, t0, prm, out); ctx.linearize_raw(R1, t1, prm, out)
    b = time.perf_counter()
    print("%-34s linearize %.1f us (n_eff %d)" % (label, (b - a) / (K // 2 * 2) * 1e6, out.n_eff), flush=True)


ctx = dcreg_amd.Context(0)
ctx.set_target(tgt, radius)
ctx.set_source(src)
measure(ctx, "library order (Hilbert)")
ctx.set_option("keep_source_order", 1)
order = np.argsort(morton(src), kind="stable")
ctx.set_source(np.ascontiguousarray(src[order]))
measure(ctx, "numpy Morton, kept")
dbg = ctx.linearize(T0[:3, :3], T0[:3, 3], prm, debug=True)     # warm statistics: call twice so that the bound is warm
dbg = ctx.linearize(T1[:3, :3], T1[:3, 3], prm, debug=True)
ev = (dbg["stats"] & 0xFFFF).astype(np.int64)                    # per point, in the order passed (= Morton)
for CH in (256, 1024, 4096):
    o2 = order.copy()
    for a in range(0, len(o2), CH):
        sl = slice(a, min(a + CH, len(o2)))
        o2[sl] = o2[sl][np.argsort(ev[sl], kind="stable")]
    ctx.set_source(np.ascontiguousarray(src[o2]))
    measure(ctx, "Morton chunks of %d sorted by work" % CH)
