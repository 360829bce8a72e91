"""Host replay: visit counters (candidates, table entries, rows) of the ring walk and of the row sweep on a misaligned corridor - fresh
search, then two warm-bounded ones (profiles/r03_ablation.md section 9).  CPU only."""
import os
import sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import emul, helpers as h
n = 150_000
L = 200.0 * n / 1e6
tgt = h.scene_corridor(n, seed=100, length=L)
rng = np.random.default_rng(1100)
src = (tgt + rng.normal(0, 0.01, tgt.shape)).astype(np.float32)
yaw = 0.87 / (L / 2)
def pose(f):
    T = h.pose6d_matrix(0.05 * f, -0.08 * f, 0.03 * f, h.deg2rad(0.2) * f, h.deg2rad(-0.1) * f, yaw * f)
    return T[:3, :3], T[:3, 3]
idx = emul.Index(tgt, 1.0)
names = ["cand", "shell", "loads", "rows", "runs", "trips", "faces", "skips"]
for sw in (0, 1):
    idx.set_sweep(sw)
    s = emul.Source(src)
    for it, f in enumerate((1.0, 0.85, 0.7)):
        R, t = pose(f)
        r = emul.linearize(idx, s, R, t, stats=True, plan="full")
        w = emul.wave_cost(r["stats"]).astype(np.float64)
        print("sweep", sw, "iter", it, "searched", r["searched"], "sum/1e3:", " ".join(f"{nm}={w[:, i].sum()/1e3:.1f}" for i, nm in enumerate(names)),
              "| p99:", " ".join(f"{nm}={np.percentile(w[:, i], 99):.0f}" for i, nm in enumerate(names)))
