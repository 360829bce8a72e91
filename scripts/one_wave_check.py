"""GPU: the one-wave instantiation of k_lin (option "one_wave") changes no sum - walks on the scenes of tests/test_gpu_round5.py, both
plane fits, with and without certificates; and an engine run of a 400 k corridor pair."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import helpers as h
from dcreg_amd import api
from test_gpu_round5 import _scene, _same_sums

bad = 0
for scene in ("cylinder_60k", "fixture", "lattice_dups", "planes_dense", "corridor_300k"):
    for fast in (1, 0):
        rng = np.random.default_rng(31)
        tgt, src, radius = _scene(scene, rng)
        prm = api.default_lin_params(radius, 1)
        ctxs = {}
        for name, opts in (("two", {"one_wave": 2}), ("two_nocert", {"one_wave": 2, "use_certificates": 0}), ("plain", {"one_wave": 0})):
            c = api.Context(0)
            c.set_option("fast_plane_fit", fast)
            for k, v in opts.items():
                c.set_option(k, v)
            c.set_target(tgt, radius); c.set_source(src)
            ctxs[name] = c
        T = np.eye(4)
        steps = [0.0, 1e-6, 1e-4, 3e-4, 1e-3, -1e-3, 2e-3, 1e-5, 4e-3, 6e-3, -6e-3, 1e-2, 1e-4, 3e-2, 0.2, 1e-3, 5e-4, 0.0]
        for k, sz in enumerate(steps):
            T = h.pose6d_matrix(sz * 0.6, -sz * 0.3, sz * 0.2, sz * 0.002, -sz * 0.001, sz * 0.004) @ T
            outs = {name: c.linearize(T[:3, :3], T[:3, 3], prm) for name, c in ctxs.items()}
            if not (_same_sums(outs["two"], outs["plain"]) and _same_sums(outs["two_nocert"], outs["plain"])):
                bad += 1; print("MISMATCH", scene, fast, k, outs["two"]["n_eff"], outs["plain"]["n_eff"])
        for c in ctxs.values():
            c.close()
        print(scene, fast, "done", flush=True)
tgt = h.scene_corridor(400_000, seed=9)
src = (tgt + np.random.default_rng(10).normal(0, 0.01, tgt.shape)).astype(np.float32)
T0 = h.pose6d_matrix(0.05, -0.08, 0.03, h.deg2rad(0.2), h.deg2rad(-0.1), h.deg2rad(0.5))
cfg = api.default_config(search_radius=1.0, max_iterations=30, KAPPA_TARGET=10.0, STD_REG_GAMMA=100.0, CONVERGENCE_THRESH_ROT=0.0,
                         CONVERGENCE_THRESH_TRANS=0.0, use_weight_derivative=1, always_compute_schur=1)
logs = {}
for name, tt in (("two", 2), ("rule", 1), ("plain", 0)):
    c = api.Context(0)
    c.set_option("one_wave", tt)
    c.set_target(tgt, 1.0); c.set_source(src)
    res, lg = c.icp_run(T0, "Ours", cfg)
    logs[name] = [(np.array(L.H_upper[:]), np.array(L.gradient[:]), L.effective_points, np.array(L.transform_matrix[:])) for L in lg[:res.iterations]]
    c.close()
for name in ("two", "rule"):
    for it, (x, y) in enumerate(zip(logs[name], logs["plain"])):
        if not (np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) and x[2] == y[2] and np.array_equal(x[3], y[3])):
            bad += 1; print("ENGINE MISMATCH", name, it)
print("one_wave_check:", "OK" if bad == 0 else "%d mismatches" % bad)
sys.exit(1 if bad else 0)
