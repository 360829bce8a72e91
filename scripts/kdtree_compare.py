"""GPU box: the grid index against the kd-tree comparator (csrc/device/kdtree.hip) on the exact 5-NN search of the bench workloads -
SURVEY.md 7.1 "benchmark both, keep whichever wins per density regime".  Same queries, same kernel shape (one thread per query, the
plain k-NN kernel: no warm bound, no certificates - what BOTH structures cost when nothing is known), HIP events around the launches;
the lists are compared bit for bit.  Writes a markdown table to stdout; usage: python scripts/kdtree_compare.py [tag]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dcreg_amd
from dcreg_amd import api, scenes as h
import bench

rows = []
for wl in ("c4_corridor_1m", "c3_pk01_200k", "c2_cylinder_100k", "c1_fixture_7562"):
    W = bench.WORKLOADS[wl]
    tgt, src = bench.make_pair(W["scene"], W["n"], seed=100)
    T0 = bench.initial_pose(W["scene"])
    ctx = dcreg_amd.Context(0)
    ctx.set_target(tgt, W["radius"]); ctx.set_source(src)
    info = ctx.index_info()
    cfg = api.default_config(search_radius=W["radius"], max_iterations=W["run_len"], KAPPA_TARGET=10.0, STD_REG_GAMMA=100.0, CONVERGENCE_THRESH_ROT=0.0,
                             CONVERGENCE_THRESH_TRANS=0.0, use_weight_derivative=W["wd"], always_compute_schur=1)
    res, _ = ctx.icp_run(T0, "Ours", cfg)
    Rc, tc = np.array(res.R[:]).reshape(3, 3), np.array(res.t[:])
    qs = {"at the converged pose of a run": (src.astype(np.float64) @ Rc.T + tc).astype(np.float32),
          "at the run's initial pose": (src.astype(np.float64) @ T0[:3, :3].T + T0[:3, 3]).astype(np.float32)}
    for leaf in (8, 16, 32):
        depth, _, build_ms = ctx.kdtree_build(leaf)
        for qname, q in qs.items():
            ig, dg, tg = ctx.knn_timed(q, 5, W["radius"], "grid", repeats=5)
            i2, d2_, ts = ctx.knn_timed(q, 5, W["radius"], "grid_sweep", repeats=5)
            ik, dk, tk = ctx.knn_timed(q, 5, W["radius"], "kdtree", repeats=5)
            same = bool(np.array_equal(ig, ik) and np.array_equal(dg.view(np.uint32), dk.view(np.uint32)) and np.array_equal(ig, i2) and np.array_equal(dg.view(np.uint32), d2_.view(np.uint32)))
            rows.append((wl, qname, leaf, depth, build_ms, 1e3 * tg, 1e3 * ts, 1e3 * tk, same))
    ctx.close()
print("| workload | queries | kd-tree leaf / depth | host build (ms) | grid, ring walk (µs) | grid, row sweep (µs) | kd-tree (µs) | kd-tree / best grid | same lists |")
print("|---|---|---|---|---|---|---|---|---|")
for r in rows:
    print("| %s | %s | %d / %d | %.0f | %.1f | %.1f | %.1f | %.2f | %s |" % (r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[7] / min(r[5], r[6]), "yes" if r[8] else "NO"))
