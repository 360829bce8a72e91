"""GPU probe (round 6): the TIMELINE of a far-off launch on ONE clock (the shader clocks of the eight XCDs are not synchronised).  Needs a
variant of the library whose k_lin<2> stamps 0 (block start) and 5 (wave end) read wall_clock64() (s_memrealtime, 10 ns ticks) instead of
the cycle counter: replace `__builtin_readcyclecounter()` by `wall_clock64()` in those two stamp() calls of kernels.hpp, build with
DCREG_BUILD_TAG=wall (dcreg_amd/build.py: lib/libdcreg_hip_wall.so), restore the file.  Pose of iteration `it` of the C4 run, cold.
usage: DCREG_LIB=dcreg_amd/lib/libdcreg_hip_wall.so timeline_probe.py [iteration]      (profiles/r06_ablation.md section 3)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, emul
from dcreg_amd import api
it = int(sys.argv[1]) if len(sys.argv) > 1 else 0
W = bench.WORKLOADS["c4_corridor_1m"]
tgt, src = bench.make_pair(W["scene"], W["n"], seed=100)
src = np.ascontiguousarray(src[emul.hilbert_order(src)])
T0 = bench.initial_pose(W["scene"])
prm = api.default_lin_params(W["radius"], W["wd"])
ctx = api.Context(0)
for kv in sys.argv[2:]:
    k, v = kv.split("="); ctx.set_option(k, float(v))
ctx.set_option("keep_source_order", 1)
ctx.set_target(tgt, W["radius"]); ctx.set_source(src)
R, t = T0[:3, :3], T0[:3, 3]
ctx.dcreg_hint = None
ctx.hint_misalignment(1.0)
ctx.linearize(R, t, prm)                      # the order is estimated at this pose
for rep in range(2):
    ctx.reset_warm_state(-1)
    ctx.hint_misalignment(1.0)
    _, st = ctx.linearize_stamped(R, t, prm)
nw = len(src) // 64
st = st[:nw].astype(np.int64)
t0 = st[:, 0].min()
start = (st[:, 0] - t0) / 100.0; end = (st[:, 5] - t0) / 100.0          # us
dur = end - start
print("launch by the stamps: %.1f us from the first block's start to the last wave's end; waves %d" % (end.max(), nw))
print("wave duration us: mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f; sum / 4096 = %.1f us" % (dur.mean(), *np.percentile(dur, [50, 90, 99, 100]), dur.sum() / 4096))
edges = [0, 1, 5, 10, 20, 40, 60, 80, 100, 120, 140, 160, 180, 200, 220, 240, 260, 300, 400]
print("started in window (us): waves, mean duration, max end")
for a, b in zip(edges[:-1], edges[1:]):
    m = (start >= a) & (start < b)
    if m.any():
        print("  [%3d, %3d): %6d waves, mean duration %6.1f us, p99 %6.1f, latest end %6.1f" % (a, b, m.sum(), dur[m].mean(), np.percentile(dur[m], 99), end[m].max()))
late = np.argsort(-end)[:15]
print("the waves that end last: wave, start, end, duration (us)")
for w in late:
    print("  %6d  %6.1f  %6.1f  %6.1f" % (w, start[w], end[w], dur[w]))
# resident waves over time
ts = np.linspace(0, end.max(), 27)
print("resident waves at t (us):", " ".join("%d@%.0f" % (int(((start <= x) & (end > x)).sum()), x) for x in ts))
