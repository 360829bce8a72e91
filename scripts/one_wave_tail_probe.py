"""GPU: what a launch of k_lin lasts in four-wave and in one-wave blocks at a FIXED pose (wall clock over 300 launches): settled
(certificates hold), every point searched from a warm state at the aligned pose, and every point searched 0.87 m off."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench
from dcreg_amd import api

w = bench.WORKLOADS["c4_corridor_1m"]
tgt, src = bench.make_pair(w["scene"], w["n"], 100)
T0 = bench.initial_pose(w["scene"])
prm = api.default_lin_params(w["radius"], w["wd"])
for label, T, cert in (("settled", np.eye(4), 1), ("all searched, aligned", np.eye(4), 0), ("all searched, 0.87 m off", T0, 0)):
    out = []
    for ow in (0, 2):
        c = api.Context(0)
        c.set_option("one_wave", ow); c.set_option("use_certificates", cert)
        c.set_target(tgt, w["radius"]); c.set_source(src)
        R, t = np.ascontiguousarray(T[:3, :3]), np.ascontiguousarray(T[:3, 3])
        for _ in range(20):
            c.linearize(R, t, prm)
        ta = time.perf_counter()
        for _ in range(300):
            c.linearize(R, t, prm)
        out.append(1e6 * (time.perf_counter() - ta) / 300)
        c.close()
    print("%-28s four-wave %7.1f us   one-wave %7.1f us   (%+.1f)" % (label, out[0], out[1], out[1] - out[0]))
