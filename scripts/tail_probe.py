"""Experiment driver (GPU box): time of the settled 1 M launch - the same pose linearised again and again, every point on its stored
plane - by HIP events, waiting with a stream synchronise ("spin" = 0) so that it also works with experiment builds that do not publish
results (DCREG_LIB=dcreg_amd/lib/libdcreg_hip_<tag>.so)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, dcreg_amd
from dcreg_amd import api
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "c4_corridor_1m"
W = bench.WORKLOADS[wl]
tgt, src = bench.make_pair(W["scene"], W["n"], seed=100)
ctx = dcreg_amd.Context(0)
ctx.set_option("spin", 0)
for kv in sys.argv[2:]:
    k, v = kv.split("="); ctx.set_option(k, float(v))
ctx.set_target(tgt, W["radius"]); ctx.set_source(src)
T = np.eye(4)
prm = api.default_lin_params(W["radius"], 1)
ctx.linearize(T[:3, :3], T[:3, 3], prm)
ctx.linearize(T[:3, :3], T[:3, 3], prm)
ctx.set_option("time_kernels", 1)
for rep in range(3):
    ctx.kernel_time(reset=True)
    for k in range(100):
        ctx.linearize(T[:3, :3], T[:3, 3], prm)
    km, kn = ctx.kernel_time(reset=True)
    print(wl, os.environ.get("DCREG_LIB", "product"), "settled launch: %.2f us (HIP events, %d launches)" % (1e3 * km / max(kn, 1), kn))
