"""CPU model (round 6): how many candidates would the shared tile of a wave hold (search.hpp tile_search6), for the waves of a C4 launch
at the poses of iterations 0..K - with the ideal radius (the largest true 6th-neighbour distance of the wave), with the radius inflated
by a few centimetres (what a probe-derived bound gives), and what the lanes need on their own.  usage: tile_model6.py [n_points] [K]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import emul, bench
from oracle import pyoracle as po
from scipy.spatial import cKDTree

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
tgt, src = bench.make_pair("corridor", n, seed=100)
T0 = bench.initial_pose("corridor")
tree = po.KdTree(tgt)
cfg = po.default_config(search_radius=1.0, max_iterations=K + 1, thresh_rot=0.0, thresh_trans=0.0, kappa_target=10.0, std_reg_gamma=100.0,
                        use_weight_derivative=1, always_compute_schur=1, num_threads=8)
res, logs = po.icp_run(tree, src, T0, "Ours", cfg)
poses = [T0] + [np.array(L.T[:]).reshape(4, 4) for L in logs[:K]]
idx = emul.Index(tgt, 1.0)
S = emul.Source(src)
kd = cKDTree(tgt.astype(np.float64))
print("cell %.4f m, %d points" % (idx.cell, n))
pc = lambda a: "mean %.0f p50 %d p90 %d p99 %d" % ((np.mean(a),) + tuple(np.percentile(a, [50, 90, 99])))
for it, T in enumerate(poses):
    q = (S.sorted.astype(np.float64) @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
    nw = len(q) // 64
    sel = np.arange(0, nw, max(1, nw // 400))
    d6 = kd.query(q[(sel[:, None] * 64 + np.arange(64)).ravel()].astype(np.float64), k=6)[0][:, 5].reshape(len(sel), 64)
    rows = []
    for wi, w in enumerate(sel):
        qq = q[w * 64:(w + 1) * 64]
        lo, hi = qq.min(0), qq.max(0)
        ext = hi - lo
        dmax, dmin = float(d6[wi].max()), float(d6[wi].min())
        out = [ext.max(), dmin, dmax]
        for r in (dmax * 1.005, dmax + 0.02, dmax + 0.05, np.sqrt(dmax * dmax + 0.1 * 0.1)):
            nt, _ = emul.tile_points(idx, lo, hi, np.float32(r * r), len(tgt), max_slots=1 << 20)
            out.append(nt)
        rows.append(out)
    a = np.array(rows, dtype=np.float64)
    far = a[:, 1] > 2 * idx.cell
    print("pose %d: waves far (min d6 > 2 cells) %.2f | box extent %s cm | d6 min %s cm max %s cm" % (it, far.mean(), pc(a[:, 0] * 100), pc(a[:, 1] * 100), pc(a[:, 2] * 100)))
    for name, sel2 in (("all", np.ones(len(a), bool)), ("far", far)):
        if sel2.any():
            print("   %-4s tile candidates: ideal (max d6 x 1.005) %s | + 2 cm %s | + 5 cm %s | in-surface + 10 cm %s" % (
                name, pc(a[sel2, 3]), pc(a[sel2, 4]), pc(a[sel2, 5]), pc(a[sel2, 6])), flush=True)
