"""GPU probe (round 6): WHICH waves of a far-off launch are the slow ones, and why.  The source in the device's curve order (option
keep_source_order, so that point i is lane i % 64 of wave i / 64), the pose of iteration `it` of the C4 run from a cold state: the debug
dump (candidates evaluated, nearest-neighbour distance per point) and the phase stamps of every wave (k_lin<2>).
usage: slow_wave_probe.py [iteration] [opts k=v ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, emul
from dcreg_amd import api

it = int(sys.argv[1]) if len(sys.argv) > 1 else 0
extra = [a.split("=") for a in sys.argv[2:]]
W = bench.WORKLOADS["c4_corridor_1m"]
tgt, src = bench.make_pair(W["scene"], W["n"], seed=100)
src = np.ascontiguousarray(src[emul.hilbert_order(src)])
T0 = bench.initial_pose(W["scene"])
prm = api.default_lin_params(W["radius"], W["wd"])
ctx = api.Context(0)
for k, v in extra:
    ctx.set_option(k, float(v))
ctx.set_option("keep_source_order", 1)
ctx.set_target(tgt, W["radius"]); ctx.set_source(src)
T = T0
if it > 0:
    cfg = api.default_config(search_radius=W["radius"], max_iterations=it, KAPPA_TARGET=10.0, STD_REG_GAMMA=100.0, CONVERGENCE_THRESH_ROT=0.0,
                             CONVERGENCE_THRESH_TRANS=0.0, use_weight_derivative=W["wd"], always_compute_schur=1)
    res, logs = ctx.icp_run(T0, "Ours", cfg)
    T = np.array(logs[it - 1].transform_matrix[:]).reshape(4, 4)
    ctx.reset_warm_state(-1)
R, t = T[:3, :3], T[:3, 3]
d = ctx.linearize(R, t, prm, debug=True)
ne = (d["stats"] & 0xFFFF).astype(np.int64)
d1 = np.sqrt(d["nn_d2"][:, 0].astype(np.float64)); d5 = np.sqrt(d["nn_d2"][:, 4].astype(np.float64))
ctx.reset_warm_state(-1)
ctx.set_option("time_kernels", 1); ctx.kernel_time(reset=True)
ctx.linearize(R, t, prm)
ms, n = ctx.kernel_time(reset=True)
ctx.set_option("time_kernels", 0)
ctx.reset_warm_state(-1)
_, st = ctx.linearize_stamped(R, t, prm)
nw = len(src) // 64
st = st[:nw]
tot = (st[:, 5] - st[:, 0]).astype(np.int64); srch = (st[:, 2] - st[:, 1]).astype(np.int64); start = (st[:, 0] - st[:, 0].min()).astype(np.int64)
end = (st[:, 5] - st[:, 0].min()).astype(np.int64)
q = (src.astype(np.float64) @ R.T + t)
print("pose of iteration %d, cold: launch %.1f us (events); waves %d; wave cycles: sum / 4096 = %.0f (%.1f us at 2.4 GHz), max %d (%.1f us), last wave ends at %d cycles (%.1f us)" % (
    it, 1e3 * ms / max(n, 1), nw, tot.sum() / 4096.0, tot.sum() / 4096.0 / 2400.0, tot.max(), tot.max() / 2400.0, end.max(), end.max() / 2400.0))
pc = lambda a: "mean %.0f p50 %d p90 %d p99 %d p99.9 %d max %d" % ((np.mean(a),) + tuple(np.percentile(a, [50, 90, 99, 99.9, 100])))
print("whole-wave cycles:", pc(tot)); print("search-phase cycles:", pc(srch))
ne_w = ne[:nw * 64].reshape(nw, 64); d1_w = d1[:nw * 64].reshape(nw, 64); d5_w = d5[:nw * 64].reshape(nw, 64)
fin = np.isfinite(d5_w)
print("per-wave max candidates:", pc(ne_w.max(1)), "| per-wave mean:", pc(ne_w.mean(1)))
print("correlation of wave cycles with: max candidates %.3f, mean candidates %.3f" % (np.corrcoef(tot, ne_w.max(1))[0, 1], np.corrcoef(tot, ne_w.mean(1))[0, 1]))
order = np.argsort(-tot)[:20]
print(" wave      cycles   start      end  cand max / mean  lanes d5=inf  d1 cm min / mean / max  x of the wave (m)  box extent (cm)")
for w in order:
    qq = q[w * 64:(w + 1) * 64]
    f = fin[w]
    dd = d1_w[w][np.isfinite(d1_w[w])] * 100
    print("%6d %10d %8d %8d   %4d / %6.1f   %3d      %s   %8.1f   %6.0f" % (w, tot[w], start[w], end[w], ne_w[w].max(), ne_w[w].mean(), int((~f).sum()),
          ("%5.1f / %5.1f / %5.1f" % (dd.min(), dd.mean(), dd.max())) if len(dd) else "  -  ", qq[:, 0].mean(), (qq.max(0) - qq.min(0)).max() * 100))
# what the launch would last if the slowest waves were as fast as the p99 wave
for cap_q in (99.9, 99.0, 95.0):
    cap = np.percentile(tot, cap_q)
    print("waves capped at the p%.1f wave (%d cycles): sum / 4096 = %.1f us" % (cap_q, cap, np.minimum(tot, cap).sum() / 4096.0 / 2400.0))
# the lanes of the slowest wave: is a lane's work explained by how far it is from its surface?
w = order[0]
print("lanes of wave %d: (d1 cm, d5 cm, candidates)" % w)
print(" ".join("(%.0f,%.0f,%d)" % (d1_w[w][l] * 100, d5_w[w][l] * 100 if np.isfinite(d5_w[w][l]) else -1, ne_w[w][l]) for l in range(64)))
# candidates against the ideal: target points inside the ball of the TRUE 6th-neighbour distance (what an oracle bound would scan at least)
