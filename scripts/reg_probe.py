"""One registration of an 8 k-point frame against the resident 200 k-point map, from host buffers (bench.py c3_registration), timed with
different options: python scripts/reg_probe.py [key=value ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dcreg_amd
from dcreg_amd import api, scenes as h
tgt, src = h.scene_parkinglot()
gt, T0 = h.pose6d_matrix(**h.PK01_GT), h.pose6d_matrix(**h.PK01_INIT)
ctx = dcreg_amd.Context(0)
for kv in sys.argv[1:]:
    k, v = kv.split("="); ctx.set_option(k, float(v))
ctx.set_target(tgt, 0.5)
cfg = api.default_config(search_radius=0.5, max_iterations=30, KAPPA_TARGET=10.0, STD_REG_GAMMA=100.0, CONVERGENCE_THRESH_ROT=1e-5,
                         CONVERGENCE_THRESH_TRANS=1e-3, use_weight_derivative=0, always_compute_schur=1, gt_matrix=gt.reshape(16))
ctx.set_option("record_launches", 1)
t_tot, t_src = [], []
for rep in range(23):
    ta = time.perf_counter(); ctx.set_source(src); tb = time.perf_counter()
    res, logs = ctx.icp_run(T0, "Ours", cfg, log_capacity=0 if rep < 22 else None)
    tc = time.perf_counter()
    if rep >= 3 and rep < 22:
        t_tot.append(tc - ta); t_src.append(tb - ta)
    if rep < 22:
        ctx.launch_series(reset=True)
ser = ctx.launch_series(reset=True)
print("registration: total %.1f us (min %.1f), set_source %.1f us, %d iterations" % (1e6 * np.mean(t_tot), 1e6 * np.min(t_tot), 1e6 * np.mean(t_src), res.iterations))
print("per-iteration us:", " ".join("%.0f" % (1e3 * L.iter_time_ms) for L in logs[:res.iterations]))
print("searched:", " ".join("%d" % x for x in ser["searched"]))
print("refitted:", " ".join("%d" % x for x in ser["refitted"]))
print("pass    :", "".join(".AT"[int(a)] for a in ser["advanced"]))
# how much of a launch's searching is left to k_lin behind the pass: the counter of "count_searches" sees k_lin's searching waves only,
# the launch's own report (launch_series) covers both kernels
ctx.set_option("count_searches", 1)
ctx.set_source(src)
info = ctx.index_info()
print("grid cell %.4f m dims %s cells %d" % (info.cell, tuple(info.dims), info.n_cells))
prm = api.default_lin_params(0.5, 0)
T = T0.copy()
ctx.linearize(T[:3, :3], T[:3, 3], prm)
ctx.launch_stats(reset=True); ctx.launch_series(reset=True)
ctx.set_option("team_stamps", 1)
Tc = np.eye(4); Tc[:3, :3] = np.array(res.R[:]).reshape(3, 3); Tc[:3, 3] = res.t[:]      # the converged pose of the last registration
for k in range(8):
    T = (h.pose6d_matrix(0.01, -0.006, 0.003, 0.0004, -0.0002, 0.0008) @ T) if k < 4 else (h.pose6d_matrix(0.002 * (8 - k), 0.001, 0.0, 0.0, 0.0, 0.0001) @ Tc)
    ctx.linearize(T[:3, :3], T[:3, 3], prm)
    st = ctx.launch_stats(reset=True); se = ctx.launch_series(reset=True)
    hist = ctx.team_pass_stamps()
    print("step %d: pass %d, launch searched %d refitted %d; k_lin's own searching lanes %d (team-searched inside k_lin %d); outcomes %s" % (
        k, se["advanced"][0], se["searched"][0], se["refitted"][0], st["points_searched"], st["points_team"], hist[-1].tolist() if len(hist) else []))
ctx.set_option("team_stamps", 0)
# where a block of the pass spends its time (shader clock, 100 MHz counter x ... : s_memtime counts shader cycles)
ctx.set_option("team_stamps", 1)
T = h.pose6d_matrix(0.01, -0.006, 0.003, 0.0004, -0.0002, 0.0008) @ T
ctx.linearize(T[:3, :3], T[:3, 3], prm)
st = ctx.team_pass_stamps().astype(np.int64)
if len(st):
    print("  outcomes [served SET, layers, wide, rows, list overflow, served OUT, served without slack, refit]:", st[-1].tolist())
    st = st[:-1]
    busy = st[:, 7] > 0
    d = np.diff(st[busy], axis=1)
    names = ["tests", "old nb", "rows", "tables", "cands", "rank", "cert+fit+store"]
    print("team pass stamps: %d blocks, %d with a round; launch span %.1f k cycles; block life mean %.1f k (p90 %.1f k)" % (
        len(st), busy.sum(), (st[:, 1:].max() - st[st[:, 0] > 0, 0].min()) / 1e3, (st[busy, 7] - st[busy, 0]).mean() / 1e3,
        np.percentile(st[busy, 7] - st[busy, 0], 90) / 1e3))
    print("  phase means (cycles): " + ", ".join("%s %.0f" % (n_, v) for n_, v in zip(names, d.mean(0))))
    print("  block start times (k cycles after the first): p10 %.1f p50 %.1f p90 %.1f max %.1f" % tuple(
        np.percentile(st[st[:, 0] > 0, 0] - st[st[:, 0] > 0, 0].min(), [10, 50, 90, 100]) / 1e3))
# the fixed cost of the extra kernel: blocking linearisations at ONE pose (nothing to search: the pass tests and leaves), pass off / forced
ctx.set_option("team_stamps", 0); ctx.set_option("count_searches", 0); ctx.set_option("record_launches", 0)
from dcreg_amd.api import LinOut
import ctypes as C
out = LinOut(); R_ = np.ascontiguousarray(T[:3, :3]).reshape(9).copy(); t_ = T[:3, 3].copy()
for mode in (0, 2, 0, 2):
    ctx.set_option("team_pass", mode)
    for _ in range(50): ctx.linearize_raw(R_, t_, prm, out)
    ta = time.perf_counter()
    for _ in range(400): ctx.linearize_raw(R_, t_, prm, out)
    print("settled pose, team_pass=%d: %.2f us per blocking linearisation" % (mode, 1e6 * (time.perf_counter() - ta) / 400))
# which points does k_lin search behind the pass?  The same pose again: nothing moved, so whatever is searched now had a certificate
# without slack (or no certificate at all)
ctx.set_option("count_searches", 1); ctx.set_option("record_launches", 1)
for mode in (2, 0):
    ctx.set_option("team_pass", mode)
    T = h.pose6d_matrix(0.01, -0.006, 0.003, 0.0004, -0.0002, 0.0008) @ T
    for rep in range(3):
        ctx.launch_stats(reset=True); ctx.launch_series(reset=True)
        ctx.linearize(T[:3, :3], T[:3, 3], prm)
        st = ctx.launch_stats(reset=True); se = ctx.launch_series(reset=True)
        print("team_pass=%d, same pose, launch %d: launch searched %d refitted %d; k_lin's own searching lanes %d" % (mode, rep, se["searched"][0], se["refitted"][0], st["points_searched"]))
