"""Model (host replay, no GPU): what the searches of a far-off pose would cost on a COARSER grid level.  Replays the first launches of a
C4 run on the product's grid and on grids with 2 / 4 / 8 times the cell edge, prints the wave-synchronous visit counters (tests/emul.py
wave_cost) for (i) every query on the fine grid (today), (ii) every query on the coarse grid, (iii) the hybrid the kernel would run:
queries whose 27-cell block on the fine grid is empty (they sweep today) on the coarse grid, the others on the fine one.
usage: python scripts/coarse_model.py [n_points] [iterations]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from dcreg_amd import scenes as h
import emul
from oracle import pyoracle as po

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4
tgt = h.scene_corridor(n, seed=100)
src = (tgt + np.random.default_rng(1100).normal(0, 0.01, tgt.shape)).astype(np.float32)
T0 = h.pose6d_matrix(0.05, -0.08, 0.03, h.deg2rad(0.2), h.deg2rad(-0.1), h.deg2rad(0.5))
tree = po.KdTree(tgt)
cfg = po.default_config(search_radius=1.0, max_iterations=50, thresh_trans=0.0, thresh_rot=0.0, kappa_target=10.0, std_reg_gamma=100.0,
                        use_weight_derivative=1, always_compute_schur=1, num_threads=8)
res, logs = po.icp_run(tree, src, T0, "Ours", cfg)
poses = [T0] + [np.array(L.T[:]).reshape(4, 4) for L in logs]
fine = emul.Index(tgt, 1.0)
print("fine cell %.4f dims %s" % (fine.cell, fine.dims), flush=True)
names = ["cand", "shell", "loads", "rows", "runs", "trips", "faces", "fskip"]
# instruction model per lane of a wave (profiles/r04_ablation.md section 13): 11 per candidate slot (4 per trip) + 25 per trip, ~30 per
# table load pair (row arithmetic), ~40 per sweep row
def instr(w):
    return 44.0 * w[:, 5] + 25.0 * w[:, 5] + 15.0 * w[:, 2] + 40.0 * w[:, 3]
levels = {}
for mult in (1, 2, 4, 8):
    idx = fine if mult == 1 else emul.Index(tgt, 1.0, cell=fine.cell * mult)
    S = emul.Source(src)
    emul.linearize(idx, S, poses[-1][:3, :3], poses[-1][:3, 3], wd=1)          # the state of the END of a run, as the bench has it
    rows = []
    for k in range(iters):
        T = poses[k]
        out = emul.linearize(idx, S, T[:3, :3], T[:3, 3], wd=1, stats=True)
        rows.append(out["stats"].astype(np.int64))
    levels[mult] = rows
    print("cell x%d (%.3f m, dims %s)" % (mult, idx.cell, idx.dims), flush=True)
for k in range(iters):
    f = levels[1][k]
    far = f[:, 3] > 0                      # queries that went through the sweep on the fine grid (rows visited)
    print("iteration %d: %.1f %% of the queries sweep on the fine grid" % (k, 100.0 * far.mean()))
    for mult in (1, 2, 4, 8):
        c = levels[mult][k]
        hyb = np.where(far[:, None], c, f)
        for label, st in (("all on x%d" % mult, c), ("hybrid x%d" % mult, hyb)):
            if mult == 1 and label.startswith("hybrid"):
                continue
            w = emul.wave_cost(st.astype(np.uint32)).astype(np.int64)
            print("  %-12s per query: %s | per wave (max lane) mean: %s | model instr/wave mean %.0f p99 %.0f" % (
                label, " ".join("%s %.1f" % (a, b) for a, b in zip(names, st.mean(0)) if a in ("cand", "loads", "rows", "trips")),
                " ".join("%s %.1f" % (a, b) for a, b in zip(names, w.mean(0)) if a in ("cand", "loads", "rows", "trips")),
                instr(w).mean(), np.percentile(instr(w), 99)), flush=True)
