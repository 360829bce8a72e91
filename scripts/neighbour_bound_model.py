"""Model (host replay, no GPU): a far-off launch in TWO phases - every G-th point of the processing order searched first (dense waves), the
others bounded by the six neighbours their sampled neighbour found (a valid upper bound of their own sixth distance: exact results, checked
here against the one-phase sums).  Prints the wave-synchronous visit counters (tests/emul.py wave_cost) and the instruction model of
scripts/coarse_model.py for the first launches of a C4 run.  Result (profiles/r05_ablation.md section 5): 0.97-1.13 x today's
instructions on the jump launch, 1.24-1.39 x on the next ones - the per-WAVE maxima hardly move.  Not built.
usage: python scripts/neighbour_bound_model.py [n_points]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from dcreg_amd import scenes as h
import emul
from oracle import pyoracle as po
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
tgt = h.scene_corridor(n, seed=100)
src = (tgt + np.random.default_rng(1100).normal(0, 0.01, tgt.shape)).astype(np.float32)
T0 = h.pose6d_matrix(0.05, -0.08, 0.03, h.deg2rad(0.2), h.deg2rad(-0.1), h.deg2rad(0.5))
tree = po.KdTree(tgt)
cfg = po.default_config(search_radius=1.0, max_iterations=50, thresh_trans=0.0, thresh_rot=0.0, kappa_target=10.0, std_reg_gamma=100.0,
                        use_weight_derivative=1, always_compute_schur=1, num_threads=8)
res, logs = po.icp_run(tree, src, T0, "Ours", cfg)
poses = [T0] + [np.array(L.T[:]).reshape(4, 4) for L in logs]
idx = emul.Index(tgt, 1.0)
def instr(w):
    return 44.0 * w[:, 5] + 25.0 * w[:, 5] + 15.0 * w[:, 2] + 40.0 * w[:, 3]
names = ["cand", "shell", "loads", "rows", "runs", "trips", "faces", "fskip"]
def show(label, st):
    w = emul.wave_cost(st.astype(np.uint32)).astype(np.int64)
    print("  %-28s per query: %s | per wave: %s | instr/wave mean %.0f, total %.3g" % (label,
        " ".join("%s %.1f" % (a, b) for a, b in zip(names, st.mean(0)) if a in ("cand", "loads", "rows", "trips")),
        " ".join("%s %.1f" % (a, b) for a, b in zip(names, w.mean(0)) if a in ("cand", "loads", "rows", "trips")),
        instr(w).mean(), instr(w).sum()), flush=True)
    return instr(w).sum()
for k in (0, 1, 2):
    S = emul.Source(src)
    prevT = poses[-1] if k == 0 else poses[k - 1]
    emul.linearize(idx, S, prevT[:3, :3], prevT[:3, 3], wd=1)
    T = poses[k]
    out0 = emul.linearize(idx, S, T[:3, :3], T[:3, 3], wd=1, stats=True)
    st0 = out0["stats"].astype(np.int64)
    print("iteration %d: searched %d" % (k, out0["searched"]))
    today = show("today", st0)
    for G in (4, 8, 16):
        Sst = S.state.copy()
        stride = S.stride
        i = np.arange(S.n)
        s = (i // G) * G + G // 2
        s = np.minimum(s, S.n - 1)
        st = S.state
        X = st[13 * stride: 13 * stride + 4 * stride].reshape(stride, 4)
        Y = st[17 * stride: 17 * stride + 2 * stride].reshape(stride, 2)
        V0 = st[0: 4 * stride].reshape(stride, 4)
        X[:S.n] = X[s]; Y[:S.n] = Y[s]
        ns = i != s
        V0[:S.n][ns, 0] = 0xFFFFFFFF
        out2 = emul.linearize(idx, S, T[:3, :3], T[:3, 3], wd=1, stats=True, plan="cert")
        st2 = out2["stats"].astype(np.int64)
        assert np.array_equal(out2["H_upper"], out0["H_upper"]), "sums differ"
        p1 = show("phase 1 (every %d-th, dense)" % G, st0[G // 2::G])
        p2 = show("phase 2 (neighbour's bound)", st2)
        print("  G=%d: searched in phase 2 %d; model total %.3g against %.3g today (%.2f)" % (G, out2["searched"], p1 + p2, today, (p1 + p2) / today), flush=True)
        S.state[:] = Sst
