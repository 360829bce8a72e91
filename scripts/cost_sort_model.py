"""Model (host replay, no GPU): the searches of an all-search launch dealt to dense waves in the order of a PREDICTED cost (sorted inside
tiles of 1536 / 6144 points), so that the lanes of a wave finish together.  Wave-synchronous visit counters (tests/emul.py wave_cost) +
the instruction model of scripts/coarse_model.py, first launches of a C4 run; predictors: the query's own cost (the unreachable ideal),
its cost in the previous launch, the population of its home cell / of its 27-cell block.  Result (profiles/r05_ablation.md section 5):
ideal 0.60-0.77 x today's instructions, every real predictor 0.94-1.07 x.  Not built.
usage: python scripts/cost_sort_model.py [n_points]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from dcreg_amd import scenes as h
import emul
from oracle import pyoracle as po
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
iters = 6
tgt = h.scene_corridor(n, seed=100)
src = (tgt + np.random.default_rng(1100).normal(0, 0.01, tgt.shape)).astype(np.float32)
T0 = h.pose6d_matrix(0.05, -0.08, 0.03, h.deg2rad(0.2), h.deg2rad(-0.1), h.deg2rad(0.5))
tree = po.KdTree(tgt)
cfg = po.default_config(search_radius=1.0, max_iterations=50, thresh_trans=0.0, thresh_rot=0.0, kappa_target=10.0, std_reg_gamma=100.0,
                        use_weight_derivative=1, always_compute_schur=1, num_threads=8)
res, logs = po.icp_run(tree, src, T0, "Ours", cfg)
poses = [T0] + [np.array(L.T[:]).reshape(4, 4) for L in logs]
idx = emul.Index(tgt, 1.0)
hc = idx.cell
org = tgt.min(0).astype(np.float64) - 1e-3
def keys(p):
    c = np.floor((p.astype(np.float64) - org) / hc).astype(np.int64)
    return (c[:, 0] * 100000 + c[:, 1]) * 100000 + c[:, 2], c
tk, _ = keys(tgt)
uk, cnt = np.unique(tk, return_counts=True)
def pop(k):
    j = np.searchsorted(uk, k); j = np.minimum(j, len(uk) - 1)
    return np.where(uk[j] == k, cnt[j], 0)
def instr_lane(s):
    return 44.0 * s[:, 5] + 25.0 * s[:, 5] + 15.0 * s[:, 2] + 40.0 * s[:, 3]
def instr(w):
    return 44.0 * w[:, 5] + 25.0 * w[:, 5] + 15.0 * w[:, 2] + 40.0 * w[:, 3]
def total(st, order):
    w = emul.wave_cost(st[order].astype(np.uint32)).astype(np.int64)
    return instr(w).sum()
def tile_sort(pred, T):
    n = len(pred)
    o = np.arange(n)
    out = []
    for a in range(0, n, T):
        seg = o[a:a + T]
        out.append(seg[np.argsort(-pred[seg], kind="stable")])
    return np.concatenate(out)
S = emul.Source(src)
emul.linearize(idx, S, poses[-1][:3, :3], poses[-1][:3, 3], wd=1)
prev = None
for k in range(iters):
    T = poses[k]
    out = emul.linearize(idx, S, T[:3, :3], T[:3, 3], wd=1, stats=True)
    st = out["stats"].astype(np.int64)
    srch = st[:, 5] > 0
    q = (S.sorted.astype(np.float64) @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
    kq, cq = keys(q)
    p_home = pop(kq).astype(np.float64)
    p27 = np.zeros(len(q))
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dz in (-1, 0, 1):
                p27 += pop(((cq[:, 0] + dx) * 100000 + cq[:, 1] + dy) * 100000 + cq[:, 2] + dz)
    own = instr_lane(st)
    base = total(st, np.arange(len(st)))
    line = "iteration %d searched %.1f %%: today %.3g |" % (k, 100.0 * out["searched"] / len(st), base)
    for T_ in (1536, 6144):
        line += " tile %d: perfect %.2f" % (T_, total(st, tile_sort(own, T_)) / base)
        if prev is not None: line += " prev %.2f" % (total(st, tile_sort(prev, T_)) / base)
        line += " home %.2f p27 %.2f |" % (total(st, tile_sort(p_home, T_)) / base, total(st, tile_sort(p27, T_)) / base)
    print(line, flush=True)
    prev = own
