"""Host replay (tests/emul.py) of the C4 run, pose by pose as the oracle's ICP run gives them: points searched / refitted per iteration and
the histogram of searching lanes per wave (profiles/r03_ablation.md section 11).  CPU only; usage: python scripts/need_hist.py [n_points]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from dcreg_amd import scenes as h
import emul
from oracle import pyoracle as po
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
tgt = h.scene_corridor(n, seed=100)
rng = np.random.default_rng(1100)
src = (tgt + rng.normal(0, 0.01, tgt.shape)).astype(np.float32)
T0 = h.pose6d_matrix(0.05, -0.08, 0.03, h.deg2rad(0.2), h.deg2rad(-0.1), h.deg2rad(0.5))
tree = po.KdTree(tgt)
cfg = po.default_config(search_radius=1.0, max_iterations=50, thresh_trans=0.0, thresh_rot=0.0, kappa_target=10.0, std_reg_gamma=100.0,
                        use_weight_derivative=1, always_compute_schur=1, num_threads=8)
res, logs = po.icp_run(tree, src, T0, "Ours", cfg)
poses = [T0] + [np.array(L.T[:]).reshape(4, 4) for L in logs]
idx = emul.Index(tgt, 1.0)
S = emul.Source(src)
emul.linearize(idx, S, poses[-1][:3, :3], poses[-1][:3, 3], wd=1)
for k in range(50):
    T = poses[k]
    out = emul.linearize(idx, S, T[:3, :3], T[:3, 3], wd=1, stats=True)
    need = out["stats"][:, 1] > 0
    pad = (-len(need)) % 64
    nw = np.concatenate([need, np.zeros(pad, bool)]).reshape(-1, 64).sum(axis=1)
    hist = [int((nw == 0).sum()), int(((nw >= 1) & (nw <= 2)).sum()), int(((nw >= 3) & (nw <= 4)).sum()), int(((nw >= 5) & (nw <= 8)).sum()), int(((nw >= 9) & (nw <= 16)).sum()), int((nw > 16).sum())]
    step = np.linalg.norm(poses[k + 1][:3, 3] - poses[k][:3, 3]) if k + 1 < len(poses) else 0
    print("iter %2d searched %7d (%.3f) fitted %7d  waves by needing lanes [0, 1-2, 3-4, 5-8, 9-16, >16] = %s  step %.2e" % (k, out["searched"], out["searched"] / len(need), out["fitted"], hist, step), flush=True)
