"""Per-point search statistics of the linearisation kernel (GPU box only)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as h
import dcreg_amd
from dcreg_amd import api
ctx = dcreg_amd.Context(0)
for name, gen, n in (("cyl100k", lambda: h.scene_cylinder(100_000, seed=1, noise=0.01), 100_000), ("corr1M", lambda: h.scene_corridor(1_000_000, seed=1), 1_000_000), ("fixture", h.cylinder_cloud, 7562)):
    tgt = gen(); rng = np.random.default_rng(0)
    src = (tgt + rng.normal(0, 0.01, tgt.shape)).astype(np.float32)
    for cf in (2.0, 1.6):
        ctx.set_option("cell_factor", cf)
        ctx.set_target(tgt, 1.0); ctx.set_source(src)
        T0 = h.pose6d_matrix(0.05, -0.08, 0.03, 0.003, -0.002, 0.008)
        T1 = h.pose6d_matrix(0.03, -0.05, 0.02, 0.002, -0.001, 0.005)      # the next ICP iterate: a slightly different pose
        for label, T in (("cold", T0), ("warm", T1)):                       # second call is bounded by the first call's neighbour sets
            out = ctx.linearize(T[:3, :3], T[:3, 3], api.default_lin_params(1.0, 1), debug=True)
            st = out["stats"]; ev = (st & 0xFFFF).astype(np.int64); sh = (st >> 16) & 0x7FFF
            wmax = ev[: len(ev) // 64 * 64].reshape(-1, 64)   # (original order: only indicative of per-wave maxima)
            print(name, "cf", cf, label, "cell %.3f" % ctx.index_info().cell, "n_eff", out["n_eff"], "eval mean %.1f p50 %d p99 %d max %d" % (ev.mean(), np.percentile(ev, 50), np.percentile(ev, 99), ev.max()),
                  "shell>1 frac %.5f" % (sh > 1).mean(), "max shell", sh.max(), flush=True)

# phase breakdown from shader-clock stamps (debug kernel), 100k cylinder
tgt = h.scene_cylinder(100_000, seed=1, noise=0.01); rng = np.random.default_rng(0)
src = (tgt + rng.normal(0, 0.01, tgt.shape)).astype(np.float32)
ctx.set_option("cell_factor", 2.0)
ctx.set_target(tgt, 1.0); ctx.set_source(src)
T0 = h.pose6d_matrix(0.05, -0.08, 0.03, 0.003, -0.002, 0.008)
T1 = h.pose6d_matrix(0.03, -0.05, 0.02, 0.002, -0.001, 0.005)
for tile in (0, 1):      # 0: cold search, 1: warm (bounded by the previous call's neighbour sets)
    ctx.set_option("warm_start", tile)
    for rep in range(2):
        out = ctx.linearize(T0[:3, :3], T0[:3, 3], api.default_lin_params(1.0, 1), debug=True)
    out = ctx.linearize(T1[:3, :3], T1[:3, 3], api.default_lin_params(1.0, 1), debug=True)
    ck = out["clocks"][: (len(src) + 63) // 64].astype(np.int64)
    c1 = np.where(ck[:, 1] > 0, ck[:, 1], ck[:, 0])
    ph = np.stack([c1 - ck[:, 0], ck[:, 2] - c1, ck[:, 3] - ck[:, 2], ck[:, 4] - ck[:, 3], ck[:, 5] - ck[:, 4], ck[:, 5] - ck[:, 0]], 1)
    names = ["tile-build", "search", "planefit+row", "wave-reduce", "block-reduce", "TOTAL"]
    pa, pb, psh = ck[:, 6] & 0xFFFFF, (ck[:, 6] >> 20) & 0xFFFFF, (ck[:, 6] >> 40) & 0xFFFFF
    print("warm", tile, "search split (lane 0 of each wave): phase A (table loads + run list) mean %d, phase B (candidates) mean %d, shells check mean %d cycles" % (pa.mean(), pb.mean(), psh.mean()))
    for k, nm in enumerate(names):
        print("   %-13s mean %7d  p50 %7d  p99 %7d  max %7d cycles" % (nm, ph[:, k].mean(), np.percentile(ph[:, k], 50), np.percentile(ph[:, k], 99), ph[:, k].max()))
