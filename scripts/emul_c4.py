"""Host replay (tests/emul.py) of the C4 workload's first iterations: per-query visit counters of the grid search and the
wave-synchronous cost model, without a GPU.  Used to design the ring walk; usage: python scripts/emul_c4.py [n_points]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dcreg_amd import scenes as h


def _rel_err(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))

import emul
from oracle import pyoracle as po

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
tgt = h.scene_corridor(n, seed=100)
rng = np.random.default_rng(1100)
src = (tgt + rng.normal(0, 0.01, tgt.shape)).astype(np.float32)
T0 = h.pose6d_matrix(0.05, -0.08, 0.03, h.deg2rad(0.2), h.deg2rad(-0.1), h.deg2rad(0.5))
t0 = time.time()
tree = po.KdTree(tgt)
cfg = po.default_config(search_radius=1.0, max_iterations=50, thresh_trans=0.0, thresh_rot=0.0, kappa_target=10.0, std_reg_gamma=100.0,
                        use_weight_derivative=1, always_compute_schur=1, num_threads=8)
res, logs = po.icp_run(tree, src, T0, "Ours", cfg)
poses = [T0] + [np.array(L.T[:]).reshape(4, 4) for L in logs]
print("oracle run %.1fs" % (time.time() - t0), flush=True)
t0 = time.time()
idx = emul.Index(tgt, 1.0, x_subdiv=int(os.environ.get("X_SUBDIV", "8")))
S = emul.Source(src)
print("index %.1fs: cell %.4f dims %s gap_cap %d" % (time.time() - t0, idx.cell, idx.dims, idx.gap_cap), flush=True)
# warm state of the END of a run (bench: every run restarts from T0 with the state of the converged pose)
emul.linearize(idx, S, poses[-1][:3, :3], poses[-1][:3, 3], wd=1)
names = ["cand", "shell", "loads", "rows", "runs", "trips", "faces", "fskip"]
for k in range(iters):
    T = poses[k]
    t0 = time.time()
    out = emul.linearize(idx, S, T[:3, :3], T[:3, 3], wd=1, stats=True)
    st = out["stats"].astype(np.int64)
    w = emul.wave_cost(out["stats"]).astype(np.int64)
    ref = logs[k]
    ok = out["n_eff"] == ref.n_eff and _rel_err(out["H_upper"], np.array(ref.H_upper[:])) < 1e-9
    print("iter %d parity %s n_eff %d | per query mean: %s | per wave max-lane mean: %s | p99 wave: %s | %.1fs" % (
        k, ok, out["n_eff"], " ".join("%s %.1f" % (a, b) for a, b in zip(names, st.mean(0))),
        " ".join("%s %.1f" % (a, b) for a, b in zip(names, w.mean(0))),
        " ".join("%s %d" % (a, b) for a, b in zip(names, np.percentile(w, 99, axis=0))), time.time() - t0), flush=True)
