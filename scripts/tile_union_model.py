"""CPU model (host replay, tests/emul.py): if the 64 queries of a wave shared ONE candidate tile in LDS - the union of the cell-table
intervals of their 27-cell blocks, cut to their search balls - how many points and intervals would it hold?  Per workload, at an
aligned pose with bounds = (distance of the 6th neighbour, inflated) as a warm search has them.
usage: tile_union_model.py [workload ...]"""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import emul, bench
from dcreg_amd import scenes as h
L = emul.lib()
L.emu_block_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
for wl in (sys.argv[1:] or ["c2_cylinder_100k", "c3_pk01_200k", "c4_corridor_1m"]):
    W = bench.WORKLOADS[wl]
    n = min(W["n"], 300_000) if wl.startswith("c4") else W["n"]
    tgt, src = bench.make_pair(W["scene"], n, seed=100)
    if wl.startswith("c4"):
        tgt, src = bench.make_pair("corridor", 1_000_000, seed=100)
        keep = tgt[:, 0] < np.percentile(tgt[:, 0], 30)            # a 30 % slice along the corridor: same density, a third of the work
        tgt, src = tgt[keep], src[keep]
    idx = emul.Index(tgt, W["radius"])
    S = emul.Source(src)
    q = S.sorted                                                 # aligned pose (identity): the settled / transition regime
    ki, kd = emul.knn(idx, q, 5, W["radius"])
    # a warm search's bound: the 6th neighbour's distance; take 1.25 x the 5th's as a stand-in, inflated by 1 %, capped at the radius
    d5 = np.where(np.isfinite(kd[:, 4]), kd[:, 4], W["radius"] ** 2)
    for moved, label in ((1.0, "warm bound (6th neighbour)"), (2.0, "bound x2 (a step of ~40 % of the spacing)")):
        b = np.minimum(d5 * 1.25 * 1.01 * moved, np.float32(W["radius"] ** 2 * 1.1)).astype(np.float32)
        rows = np.zeros((len(q), 9, 2), np.uint32)
        L.emu_block_rows(idx.ptr, q.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), len(q), rows.ctypes.data_as(C.c_void_p))
        nw = len(q) // 64
        per_lane = (rows[:, :, 1] - rows[:, :, 0]).sum(axis=1)
        tile_pts, tile_iv, lane_max = [], [], []
        for w in range(0, nw, max(1, nw // 3000)):
            r = rows[w * 64:(w + 1) * 64].reshape(-1, 2).astype(np.int64)
            r = r[r[:, 1] > r[:, 0]]
            if len(r) == 0:
                tile_pts.append(0); tile_iv.append(0); lane_max.append(0); continue
            r = r[np.argsort(r[:, 0])]
            tot, iv, cs, ce = 0, 0, r[0, 0], r[0, 1]
            for s_, e_ in r[1:]:
                if s_ <= ce: ce = max(ce, e_)
                else: tot += ce - cs; iv += 1; cs, ce = s_, e_
            tot += ce - cs; iv += 1
            tile_pts.append(tot); tile_iv.append(iv); lane_max.append(per_lane[w * 64:(w + 1) * 64].max())
        tp, ti, lm = np.array(tile_pts), np.array(tile_iv), np.array(lane_max)
        pc = lambda a: "p50 %d p90 %d p99 %d max %d" % tuple(np.percentile(a, [50, 90, 99, 100]))
        print("%s, cell %.3f m, %s: candidates per lane mean %.1f (wave max: %s) | union tile points: %s | intervals: %s | fits 256 pts: %.3f, 384: %.3f, 512: %.3f" % (
            wl, idx.cell, label, per_lane.mean(), pc(lm), pc(tp), pc(ti), (tp <= 256).mean(), (tp <= 384).mean(), (tp <= 512).mean()), flush=True)
