"""Where do C4's first iterations spend their time?  (GPU box.)  Replays the pose sequence of a run with the instrumented kernel
(dcreg_linearize_debug): per-point search statistics and per-wave shader-clock stamps, incl. the ring-walk breakdown of lane 0
(cycles inside candidate scans / waiting for table entries, row iterations, scans).  usage: c4_first_iters.py [key=value ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as h
import dcreg_amd
from dcreg_amd import api
import bench

W = bench.WORKLOADS["c4_corridor_1m"]; scene, n_pts, radius, run_len = W["scene"], W["n"], W["radius"], W["run_len"]
tgt, src = bench.make_pair(scene, n_pts, seed=100)
ctx = dcreg_amd.Context(0)
for kv in sys.argv[1:]:
    k, v = kv.split("="); ctx.set_option(k, float(v))
ctx.set_target(tgt, radius); ctx.set_source(src)
cfg = api.default_config(search_radius=radius, max_iterations=50, KAPPA_TARGET=10.0, STD_REG_GAMMA=100.0,
                         CONVERGENCE_THRESH_ROT=0.0, CONVERGENCE_THRESH_TRANS=0.0, use_weight_derivative=1, always_compute_schur=1)
T_init = bench.initial_pose(scene)
res, logs = ctx.icp_run(T_init, "Ours", cfg)          # leaves the warm state of the converged pose, as between two bench runs
poses = [T_init] + [np.array(L.transform_matrix[:]).reshape(4, 4) for L in logs]
prm = api.default_lin_params(radius, 1)
for k, T in enumerate(poses[:5]):
    out = ctx.linearize(T[:3, :3], T[:3, 3], prm, debug=True)
    st = out["stats"]; ev = (st & 0xFFFF).astype(np.int64); sh = (st >> 16) & 0x7FFF
    ck = out["clocks"][: (len(src) + 63) // 64].astype(np.int64)
    search = ck[:, 2] - np.where(ck[:, 1] > 0, ck[:, 1], ck[:, 0]); total = ck[:, 5] - ck[:, 0]
    pa, pb, psh = ck[:, 6] & 0xFFFFF, (ck[:, 6] >> 20) & 0xFFFFF, (ck[:, 6] >> 40) & 0xFFFFF
    rings = search - pa - pb
    scan, wait, rows, scans = ck[:, 8], ck[:, 9], ck[:, 10], ck[:, 11]
    ph = [ck[:, 1] - ck[:, 0], ck[:, 2] - ck[:, 1], ck[:, 3] - ck[:, 2], ck[:, 4] - ck[:, 3], ck[:, 5] - ck[:, 4]]
    print("        phases (mean cycles per wave): prologue+warm gathers %d | search %d | plane fit + row %d | wave reduction %d | block reduction + ticket %d" % tuple(x.mean() for x in ph))
    heavy = total >= np.percentile(total, 99)
    f = lambda a: "%d/%d" % (a.mean(), a[heavy].mean())
    print("iter %d: cand mean %.0f p99 %d | rings>1 %.2f | wave cycles mean/p99-waves: total %s search %s (A %s B %s rings %s) | ring walk of lane 0: "
          "scan cycles %s, table-wait cycles %s, row iterations %s, scans %s" % (
              k, ev.mean(), np.percentile(ev, 99), (sh > 1).mean(), f(total), f(search), f(pa), f(pb), f(rings), f(scan), f(wait), f(rows), f(scans)), flush=True)
