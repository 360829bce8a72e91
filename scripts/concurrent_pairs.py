"""Throughput of P independent scan pairs running CONCURRENTLY on one GPU (one context + stream + host thread per pair):
at 100 k points one linearisation kernel occupies 1.5 of the 4 waves/SIMD the register budget allows and the device idles
during every host step, so independent pairs interleave.  (GPU box only.)  usage: concurrent_pairs.py [workload] [P ...]"""
import ctypes as C
import os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dcreg_amd import scenes as h
import dcreg_amd
from dcreg_amd import api
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else "c2_cylinder_100k"
Ps = [int(v) for v in sys.argv[2:]] or [1, 2, 3, 4]
W = bench.WORKLOADS[wl]; scene, n_pts, radius, run_len = W["scene"], W["n"], W["radius"], W["run_len"]
L = api.load()
dp = C.POINTER(C.c_double)
T_init = h.pose6d_matrix(0.05, -0.08, 0.03, h.deg2rad(0.2), h.deg2rad(-0.1), h.deg2rad(0.5))
R0 = np.ascontiguousarray(T_init[:3, :3]).reshape(9).copy(); t0 = T_init[:3, 3].copy()
det, hand = api.METHODS["Ours"]
STEPS = 400 if n_pts <= 200_000 else 100

ctxs = []
for p in range(max(Ps)):
    tgt, src = bench.make_pair(scene, n_pts, seed=100 + p)
    c = dcreg_amd.Context(0)
    c.set_target(tgt, radius); c.set_source(src)
    ctxs.append(c)


def worker(c, k, barrier, out, i):
    cfg = api.default_config(search_radius=radius, max_iterations=run_len, KAPPA_TARGET=10.0, STD_REG_GAMMA=100.0,
                             CONVERGENCE_THRESH_ROT=0.0, CONVERGENCE_THRESH_TRANS=0.0, use_weight_derivative=1, always_compute_schur=1)
    res = api.IcpResult()
    barrier.wait()
    left = k
    while left > 0:
        n = min(run_len, left)
        cfg.max_iterations = n
        rc = L.dcreg_icp_run(c._h, R0.ctypes.data_as(dp), t0.ctypes.data_as(dp), api.DETECTION[det], api.HANDLING[hand], C.byref(cfg), None, 0, C.byref(res))
        assert rc == 0 and res.iterations == n
        left -= n
    out[i] = np.array(res.t[:])


for P in Ps:
    for phase, k in (("warmup", 40), ("timed", STEPS)):
        barrier = threading.Barrier(P + 1)
        out = [None] * P
        th = [threading.Thread(target=worker, args=(ctxs[i], k, barrier, out, i)) for i in range(P)]
        for t in th: t.start()
        barrier.wait()
        a = time.perf_counter()
        for t in th: t.join()
        b = time.perf_counter()
    print("%s: P=%d concurrent pairs: %.0f ICP iterations/s aggregate (%.1f us per iteration per pair)" % (wl, P, P * STEPS / (b - a), (b - a) / STEPS * 1e6), flush=True)
