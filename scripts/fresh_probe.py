"""GPU box: the first linearisation of a scan pair (nothing known: every point searched from the search radius) at an aligned pose and
at the run's initial pose, then two 50-iteration runs.  usage: python scripts/fresh_probe.py <workload> [option=value ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dcreg_amd
from dcreg_amd import api
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "c4_corridor_1m"
W = bench.WORKLOADS[wl]
tgt, src = bench.make_pair(W["scene"], W["n"], seed=100)
T0 = bench.initial_pose(W["scene"])
prm = api.default_lin_params(W["radius"], W["wd"])
for name, T in (("aligned", np.eye(4) if W["scene"] != "parkinglot" else T0), ("initial pose", T0)):
    ctx = dcreg_amd.Context(0)
    for kv in sys.argv[2:]:
        k, v = kv.split("="); ctx.set_option(k, float(v))
    ctx.set_option("time_kernels", 1)
    ctx.set_target(tgt, W["radius"]); ctx.set_source(src)
    ctx.kernel_time(reset=True)
    ctx.linearize(T[:3, :3], T[:3, 3], prm)
    ms, n = ctx.kernel_time()
    print("%s: fresh linearisation at the %s: %.1f us" % (wl, name, 1e3 * ms / max(n, 1)))
    ctx.close()
