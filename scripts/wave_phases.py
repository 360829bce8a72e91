"""Where the waves of a linearisation spend their time (GPU box): the poses of a bench run, replayed one blocking launch at a time with
the timing probe of dcreg_debug.h (dcreg_lin_debug::stamps: shader-clock stamps at the phase boundaries of k_lin, written by lane 0 of
every wave).  Per iteration: launch-level numbers and the wave-level phase durations (cycles of the 100 MHz-independent shader clock ->
microseconds at the clock the stamps imply), split by what the wave had to do.
usage: wave_phases.py [workload] [key=value options ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dcreg_amd
from dcreg_amd import api
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "c4_corridor_1m"
W = bench.WORKLOADS[wl]; scene, n_pts, radius, run_len = W["scene"], W["n"], W["radius"], W["run_len"]
tgt, src = bench.make_pair(scene, n_pts, seed=100)
ctx = dcreg_amd.Context(0)
for kv in sys.argv[2:]:
    k, v = kv.split("="); ctx.set_option(k, float(v))
ctx.set_target(tgt, radius); ctx.set_source(src)
cfg = api.default_config(search_radius=radius, max_iterations=run_len, KAPPA_TARGET=10.0, STD_REG_GAMMA=100.0,
                         CONVERGENCE_THRESH_ROT=0.0, CONVERGENCE_THRESH_TRANS=0.0, use_weight_derivative=W["wd"], always_compute_schur=1)
T_init = bench.initial_pose(scene)
res, logs = ctx.icp_run(T_init, "Ours", cfg)
poses = [T_init] + [np.array(L.transform_matrix[:]).reshape(4, 4) for L in logs[:-1]]
prm = api.default_lin_params(radius, W["wd"])
# the state is the converged run's: replay the run from its start like the second run of the bench loop does
CLK = 2.1e3      # cycles per microsecond: fitted below from the launch time (events) vs the stamps' span
ctx.set_option("time_kernels", 1); ctx.set_option("record_launches", 1)
print("%-4s %9s %9s %8s | %s" % ("iter", "searched", "refitted", "us(ev)", "waves: class count  mean cycles per phase [load, search, fit, row, reduce] total"))
for k, T in enumerate(poses):
    ctx.launch_series(reset=True)
    out, st = ctx.linearize_stamped(T[:3, :3], T[:3, 3], prm)
    ser = ctx.launch_series(reset=True)
    st = st.astype(np.int64)
    t0, t1, t2, t3, t4, t5, ns, nr = [st[:, j] for j in range(8)]
    live = t0 > 0
    heavy = live & (t2 > 0)
    span = (t5[live].max() - t0[live].min())
    line = "%-4d %9d %9d %8.1f | span %d cyc" % (k, ser["searched"][0], ser["refitted"][0], 1e3 * ser["ms"][0], span)
    print(line)
    def cls(mask, name):
        if not mask.any():
            return
        load = (t1 - t0)[mask]
        if name == "clean":
            tot = (t5 - t0)[mask]
            print("       %-22s %6d  load %6.0f  rest %6.0f  total %6.0f (p50 %6.0f p99 %6.0f)" % (name, mask.sum(), load.mean(), (tot - load).mean(), tot.mean(), np.median(tot), np.percentile(tot, 99)))
            return
        se, fi, ro, rd, tot = (t2 - t1)[mask], (t3 - t2)[mask], (t4 - t3)[mask], (t5 - t4)[mask], (t5 - t0)[mask]
        print("       %-22s %6d  load %6.0f  search %6.0f  fit %6.0f  row %6.0f  reduce %5.0f  total %6.0f (p50 %6.0f p99 %6.0f)" % (
            name, mask.sum(), load.mean(), se.mean(), fi.mean(), ro.mean(), rd.mean(), tot.mean(), np.median(tot), np.percentile(tot, 99)))
    cls(live & ~heavy, "clean")
    cls(heavy & (ns == 0), "refit only")
    cls(heavy & (ns >= 1) & (ns <= 2), "search 1-2")
    cls(heavy & (ns >= 3) & (ns <= 7), "search 3-7")
    cls(heavy & (ns >= 8) & (ns <= 32), "search 8-32")
    cls(heavy & (ns > 32), "search > 32")
