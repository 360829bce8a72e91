"""Search statistics at the poses of C4's first iterations (GPU box): what do the expensive iterations spend their time on?"""
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
ctx.set_target(tgt, radius); ctx.set_source(src)
cfg = api.default_config(search_radius=radius, max_iterations=6, KAPPA_TARGET=10.0, STD_REG_GAMMA=100.0,
                         CONVERGENCE_THRESH_ROT=0.0, CONVERGENCE_THRESH_TRANS=0.0, use_weight_derivative=1, always_compute_schur=1)
T_init = h.pose6d_matrix(0.05, -0.08, 0.03, h.deg2rad(0.2), h.deg2rad(-0.1), h.deg2rad(0.5))
res, logs = ctx.icp_run(T_init, "Ours", cfg)
poses = [T_init] + [np.array(L.transform_matrix[:]).reshape(4, 4) for L in logs]
prm = api.default_lin_params(radius, 1)
ctx.set_source(src)      # reset the warm state, then replay the pose sequence with per-point statistics
for k, T in enumerate(poses[:5]):
    out = ctx.linearize(T[:3, :3], T[:3, 3], prm, debug=True)
    st = out["stats"]; ev = (st & 0xFFFF).astype(np.int64); sh = (st >> 16) & 0x7FFF
    ck = out["clocks"][: (len(src) + 63) // 64].astype(np.int64)
    search = ck[:, 2] - np.where(ck[:, 1] > 0, ck[:, 1], ck[:, 0]); total = ck[:, 5] - ck[:, 0]
    pa, pb, psh = ck[:, 6] & 0xFFFFF, (ck[:, 6] >> 20) & 0xFFFFF, (ck[:, 6] >> 40) & 0xFFFFF
    print("iter %d: eval mean %.0f p50 %d p99 %d max %d | shells>1 %.3f, mean outer shell %.2f max %d | wave cycles: total mean %d p99 %d, search %d (centre block A %d B %d, rings %d)" % (
        k, ev.mean(), np.percentile(ev, 50), np.percentile(ev, 99), ev.max(), (sh > 1).mean(), sh.mean(), sh.max(),
        total.mean(), np.percentile(total, 99), search.mean(), pa.mean(), pb.mean(), (search - pa - pb).mean()), flush=True)
