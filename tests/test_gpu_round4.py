"""GPU tests added in round 4 (run with -m gpu on an MI355X; everything through the C-ABI):
  * the neighbour states are keyed by the parameters their contents depend on;
  * the wave-cooperative search of sparse waves (search.hpp team_search6) and the launch log;
  * a bounded randomised parity hunt (scripts/fuzz_parity.py / fuzz_engine.py, fixed seeds) with both plane fits;
  * the Monte-Carlo engine with a full batch of 256 slots changing hands, and trials of the 5000-trial run against single runs and
    the oracle;
  * the order in which the source cloud is processed only schedules."""
import numpy as np
import pytest

import helpers as h
from dcreg_amd import api
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


def _same_sums(a, b):
    return (a["n_eff"] == b["n_eff"] and a["n_pt"] == b["n_pt"] and np.array_equal(a["H_upper"], b["H_upper"]) and np.array_equal(a["g"], b["g"])
            and a["sum_r2"] == b["sum_r2"] and a["sum_b2"] == b["sum_b2"])


def test_states_are_keyed_by_the_parameters_they_depend_on():
    """A context that linearises with search radius A, then B (or other plane thresholds, or the other plane fit) must not reuse
    certificates, gate bits or planes measured against A: each call equals the same call on a fresh context, bit for bit."""
    tgt = h.scene_cylinder(30_000, seed=3, noise=0.01)
    src = (tgt[::2] + np.random.default_rng(4).normal(0, 0.01, tgt[::2].shape)).astype(np.float32)
    T = h.pose6d_matrix(0.02, -0.01, 0.015, h.deg2rad(0.1), 0.0, h.deg2rad(-0.2))
    c = api.Context(0)
    c.set_target(tgt, 2.0); c.set_source(src)

    def fresh(prm, fast=1):
        f = api.Context(0)
        f.set_option("fast_plane_fit", fast)
        f.set_target(tgt, 2.0); f.set_source(src)
        out = f.linearize(T[:3, :3], T[:3, 3], prm)
        f.close()
        return out

    pa, pb = api.default_lin_params(1.0, 1), api.default_lin_params(2.0, 1)
    a1 = c.linearize(T[:3, :3], T[:3, 3], pa)
    a2 = c.linearize(T[:3, :3], T[:3, 3], pa)                   # (same key: the state is used - and gives the same sums)
    b1 = c.linearize(T[:3, :3], T[:3, 3], pb)                   # points certified OUT at R = 1 may be in at R = 2
    a3 = c.linearize(T[:3, :3], T[:3, 3], pa)
    assert _same_sums(a1, a2) and _same_sums(a1, a3)
    assert _same_sums(a1, fresh(pa)) and _same_sums(b1, fresh(pb))
    assert b1["n_pt"] > a1["n_pt"]
    pc = api.default_lin_params(1.0, 1)
    pc.max_plane_thickness_sq = 0.02 * 0.02                      # the stored gate bits were taken against 0.2^2
    c1 = c.linearize(T[:3, :3], T[:3, 3], pc)
    assert _same_sums(c1, fresh(pc)) and c1["n_eff"] < a1["n_eff"]
    c.set_option("fast_plane_fit", 0)                           # the stored planes came from the other fit
    d1 = c.linearize(T[:3, :3], T[:3, 3], pa)
    assert _same_sums(d1, fresh(pa, fast=0))
    # batched states likewise
    c.set_option("fast_plane_fit", 1)
    c.reserve_warm_states(2)
    Rs, ts = [T[:3, :3], T[:3, :3]], [T[:3, 3], T[:3, 3]]
    x = c.linearize_batch_warm(Rs, ts, [0, 1], pa)
    y = c.linearize_batch_warm(Rs, ts, [0, 1], pb)
    z = c.linearize_batch_warm(Rs, ts, [0, 1], pa)
    assert all(np.array_equal(u["H_upper"], a1["H_upper"]) and u["n_eff"] == a1["n_eff"] for u in x + z)
    assert all(np.array_equal(u["H_upper"], b1["H_upper"]) and u["n_eff"] == b1["n_eff"] for u in y)
    c.close()


@pytest.mark.parametrize("scene", ["cylinder_60k", "fixture", "lattice_dups", "planes_dense"])
def test_team_search_of_sparse_waves_is_invisible(scene):
    """Waves with at most seven lanes to search serve them with all 64 lanes (search.hpp team_search6) instead of searching in
    lock-step.  Walks of small steps (a few lanes per wave lose their certificate) on scenes with ties, duplicates, OUT points and
    dense cells: with the team on and off, and with certificates off (every point searched in lock-step), all sums agree bit for
    bit; the team did take searches; the launch log's counts are the counters' counts."""
    rng = np.random.default_rng(12)
    if scene == "cylinder_60k":
        tgt, radius = h.scene_cylinder(60_000, seed=8, noise=0.01), 1.0
    elif scene == "fixture":
        tgt, radius = h.cylinder_cloud(), 1.0
    elif scene == "planes_dense":
        tgt, radius = h.scene_planes(80_000, seed=4), 0.4
    else:
        g = np.arange(0, 14, dtype=np.float32) * 0.3
        tgt = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
        tgt, radius = np.concatenate([tgt, tgt[::7]]), 0.7
    src = (tgt[::2] + rng.normal(0, 0.004, tgt[::2].shape)).astype(np.float32)
    if scene == "lattice_dups":
        src = (tgt[::3] + np.float32(0.11)).astype(np.float32)
    prm = api.default_lin_params(radius, 1)
    ctxs = {}
    for name, opts in (("team", {}), ("lock", {"team_search": 0}), ("all", {"use_certificates": 0})):
        c = api.Context(0)
        for k, v in opts.items():
            c.set_option(k, v)
        c.set_option("count_searches", 1); c.set_option("record_launches", 1)
        c.set_option("team_pass", 0); c.set_option("advance", 0)        # (this test is about the searches INSIDE the linearisation kernel)
        c.set_target(tgt, radius); c.set_source(src)
        ctxs[name] = c
    T = np.eye(4)
    steps = [0.0, 1e-6, 1e-4, 3e-4, 1e-3, -1e-3, 2e-3, 1e-5, 4e-3, 6e-3, -6e-3, 1e-2, 1e-4, 3e-2, 1e-3, 5e-4]
    team_pts = 0
    for k, sz in enumerate(steps):
        T = h.pose6d_matrix(sz * 0.6, -sz * 0.3, sz * 0.2, sz * 0.002, -sz * 0.001, sz * 0.004) @ T
        outs = {name: c.linearize(T[:3, :3], T[:3, 3], prm) for name, c in ctxs.items()}
        assert _same_sums(outs["team"], outs["lock"]) and _same_sums(outs["team"], outs["all"]), (scene, k)
        st = {name: c.launch_stats(reset=True) for name, c in ctxs.items()}
        assert st["lock"]["points_team"] == 0 and st["all"]["points_team"] == 0
        assert st["all"]["points_searched"] == len(src)
        # the team's seventh-neighbour bound is exact, the lock-step search's a lower bound of it: never more searches
        team_pts += st["team"]["points_team"]
        ser = ctxs["team"].launch_series(reset=True)
        assert len(ser["ms"]) == 1 and ser["searched"][0] == st["team"]["points_searched"] and ser["points"][0] == len(src)
        assert 0 <= ser["refitted"][0] <= len(src) - ser["searched"][0]
    assert team_pts > 0, scene
    for c in ctxs.values():
        c.close()


def test_bounded_parity_hunt_with_both_plane_fits():
    """scripts/fuzz_parity.py with fixed seeds inside the suite: 64 random scenes (kind, size, noise, outliers, radius, cell size,
    with / without the empty-space field and warm start) x 4 random poses from millimetres to half a metre, each with the default
    plane fit and the Eigen-shaped one: neighbour indices, float distances (bit patterns) and gate flags == oracle, N_eff / n_pt
    exact, H / g to 1e-8."""
    rng = np.random.default_rng(20260927)
    ctxs = {fast: api.Context(0) for fast in (1, 0)}
    for fast, c in ctxs.items():
        c.set_option("fast_plane_fit", fast)
    bad = []
    for case in range(64):
        kind = int(rng.integers(0, 4))
        n = int(rng.choice([800, 3000, 9000, 20000]))
        seed = int(rng.integers(1 << 30))
        if kind == 0:
            tgt = h.scene_cylinder(n, seed=seed, noise=float(rng.choice([0.0, 0.01, 0.05])))
        elif kind == 1:
            tgt = h.scene_corridor(n, seed=seed, length=float(rng.choice([20.0, 60.0])))
        elif kind == 2:
            tgt = h.scene_planes(n, seed=seed)
        else:
            tgt = (rng.uniform(-3, 3, (n, 3)) * np.array([1.0, 1.0, float(rng.choice([0.02, 1.0]))])).astype(np.float32)
        m = int(rng.integers(200, 1500))
        src = tgt[rng.integers(0, len(tgt), m)] + rng.normal(0, float(rng.choice([0.0, 0.01, 0.2])), (m, 3))
        if rng.random() < 0.3:
            src = np.concatenate([src, rng.uniform(-60, 60, (50, 3))])          # far outliers
        src = src.astype(np.float32)
        radius = float(rng.choice([0.3, 0.5, 1.0, 2.0]))
        opts = {"cell_factor": float(rng.choice([1.0, 1.5, 2.0, 3.0])), "gap_field": int(rng.integers(0, 2)), "warm_start": int(rng.integers(0, 2))}
        tree = po.KdTree(tgt)
        wd = int(rng.integers(0, 2))
        for c in ctxs.values():
            for k, v in opts.items():
                c.set_option(k, v)
            c.set_target(tgt, radius); c.set_source(src)
        for step in range(4):
            amp = float(rng.choice([0.005, 0.05, 0.5]))
            T = h.pose6d_matrix(*(rng.normal(0, amp, 3)), *(rng.normal(0, amp * 0.05, 3)))
            r = po.linearize(tree, src, T[:3, :3], T[:3, 3], po.default_lin_params(radius, wd), debug=True)
            ok = r["flag"] != 0
            for fast, c in ctxs.items():
                g = c.linearize(T[:3, :3], T[:3, 3], api.default_lin_params(radius, wd), debug=True)
                why = []
                if not np.array_equal(g["flag"], r["flag"]):
                    why.append("flags (%d differ)" % int((g["flag"] != r["flag"]).sum()))
                if not (np.array_equal(g["nn_idx"][ok], r["nn_idx"][ok]) and np.array_equal(g["nn_d2"][ok].view(np.uint32), r["nn_d2"][ok].view(np.uint32))):
                    why.append("neighbour lists")
                if g["n_eff"] != r["n_eff"] or g["n_pt"] != r["n_pt"]:
                    why.append("counts %d/%d %d/%d" % (g["n_eff"], r["n_eff"], g["n_pt"], r["n_pt"]))
                # H: the default fit agrees with the oracle's to 1e-8 here; the Eigen-shaped one multiplies by ONE reciprocal per Householder
                # step where Eigen divides (search.hpp householder_step), an ulp that the conditioning of [q_j] x = -1 tens of metres from
                # the origin amplifies
                if not why and r["n_eff"] > 0:
                    eh, eg = h.rel_err(g["H_upper"], r["H_upper"]), h.rel_err(g["g"], r["g"])
                    if not (eh < (1e-8 if fast else 1e-6) and eg < (1e-7 if fast else 1e-5)):
                        why.append("H %.2e g %.2e" % (eh, eg))
                plain = c.linearize(T[:3, :3], T[:3, 3], api.default_lin_params(radius, wd))       # (certificates in use)
                if plain["n_eff"] != r["n_eff"] or not np.array_equal(plain["H_upper"], g["H_upper"]):
                    why.append("plain launch != debug launch")
                if why:
                    bad.append((case, step, fast, kind, len(tgt), len(src), radius, "; ".join(why)))
    for c in ctxs.values():
        c.close()
    assert not bad, "\n".join(str(b) for b in bad[:24])


def test_bounded_engine_hunt():
    """scripts/fuzz_engine.py inside the suite: full ICP runs of random small scenes through dcreg_icp_run (pipelined launches,
    certificates, team search) against the oracle's runs: iteration counts, convergence, N_eff per iteration, final pose."""
    rng = np.random.default_rng(4242)
    c = api.Context(0)
    for case in range(12):
        n = int(rng.choice([2000, 6000, 15000]))
        kind = int(rng.integers(0, 3))
        seed = int(rng.integers(1 << 30))
        tgt = (h.scene_cylinder(n, seed=seed, noise=0.01) if kind == 0 else h.scene_corridor(n, seed=seed, length=30.0) if kind == 1 else h.scene_planes(n, seed=seed))
        src = (tgt[rng.integers(0, len(tgt), n // 2)] + rng.normal(0, 0.01, (n // 2, 3))).astype(np.float32)
        radius = float(rng.choice([0.5, 1.0]))
        method = ["Ours", "ME-SR", "ME-TReg", "FCN-SR"][int(rng.integers(0, 4))]
        wd = int(rng.integers(0, 2))
        amp = float(rng.choice([0.02, 0.1]))
        T0 = h.pose6d_matrix(*(rng.normal(0, amp, 3)), *(rng.normal(0, amp * 0.1, 3)))
        kw = dict(search_radius=radius, max_iterations=25, KAPPA_TARGET=10.0, STD_REG_GAMMA=100.0, use_weight_derivative=wd, always_compute_schur=1)
        cfg = api.default_config(CONVERGENCE_THRESH_TRANS=1e-4, CONVERGENCE_THRESH_ROT=1e-5, **kw)
        ocfg = po.default_config(search_radius=radius, max_iterations=25, thresh_trans=1e-4, thresh_rot=1e-5, kappa_target=10.0, std_reg_gamma=100.0,
                                 use_weight_derivative=wd, always_compute_schur=1)
        c.set_target(tgt, radius); c.set_source(src)
        res, logs = c.icp_run(T0, method, cfg)
        ores, ologs = po.icp_run(po.KdTree(tgt), src, T0, method, ocfg)
        assert res.iterations == ores.iterations and res.converged == ores.converged and res.status == ores.status, (case, method)
        for L, O in zip(logs[:res.iterations], ologs):
            assert L.effective_points == O.n_eff and L.corr_pt_count == O.n_pt, (case, method, L.iter_count)
        assert np.allclose(res.R[:], ores.R[:], atol=1e-6) and np.allclose(res.t[:], ores.t[:], atol=1e-6), (case, method)
    c.close()


def _mc_cfg(max_iterations=30):
    return api.default_config(search_radius=1.0, max_iterations=max_iterations, CONVERGENCE_THRESH_TRANS=1e-3, CONVERGENCE_THRESH_ROT=1e-5,
                              KAPPA_TARGET=10.0, STD_REG_GAMMA=100.0, DEGENERACY_THRES_COND=10.0, DEGENERACY_THRES_EIG=120.0,
                              use_weight_derivative=1, always_compute_schur=1)


def test_montecarlo_with_a_full_batch_of_256_slots():
    """640 trials through 256 slots (the batch stays full while slots - and their neighbour states - change hands 384 times) give,
    bit for bit, the records of the same trials through 64 slots and through 7; 16 of them equal single runs of their poses."""
    from dcreg_amd import montecarlo as mc
    pts = h.cylinder_cloud()
    c = api.Context(0)
    c.set_target(pts, 1.0); c.set_source(pts)
    cfg = _mc_cfg(20)
    base = (0.2, 0.8, 0.5, h.deg2rad(0.1), h.deg2rad(0.1), h.deg2rad(2.0))
    n, seed, ta, ra = 640, 31, 0.5, h.deg2rad(2.0)
    cols = [k for k in range(mc.REC) if k != mc.R_TIME]
    ref, stats = mc.run_montecarlo_native(c, "Ours", cfg, base, n, seed, ta, ra, slots=256)
    for slots in (64, 7):
        recs, _ = mc.run_montecarlo_native(c, "Ours", cfg, base, n, seed, ta, ra, slots=slots)
        assert np.array_equal(recs[:, cols], ref[:, cols]), slots
    assert len(set(ref[:, mc.R_ITERS])) > 4 and 0 < stats["converged_runs"] < n
    for k in np.random.default_rng(1).choice(n, 16, replace=False):
        res, _ = c.icp_run(mc.trial_pose(base, int(k), seed, ta, ra), "Ours", cfg)
        T = ref[k, mc.R_T:mc.R_T + 16].reshape(4, 4)
        assert ref[k, mc.R_ITERS] == res.iterations and ref[k, mc.R_CONV] == res.converged
        assert np.array_equal(T[:3, :3].reshape(9), np.array(res.R[:])) and np.array_equal(T[:3, 3], np.array(res.t[:]))
    c.close()


def test_trials_of_the_5000_trial_run_against_single_runs_and_the_oracle():
    """BASELINE config 5 as bench.py runs it (5000 trials, 256 slots, seed 2024, +-0.5 m / +-2 deg, <= 30 iterations, thresholds on):
    16 random trials are bitwise the single dcreg_icp_run of their pose and agree with the oracle's run of it (iterations, convergence,
    pose to 1e-7)."""
    from dcreg_amd import montecarlo as mc
    pts = h.cylinder_cloud()
    c = api.Context(0)
    c.set_target(pts, 1.0); c.set_source(pts)
    cfg = _mc_cfg(30)
    base = (0.2, 0.8, 0.5, h.deg2rad(0.1), h.deg2rad(0.1), h.deg2rad(2.0))
    n, seed, ta, ra = 5000, 2024, 0.5, np.deg2rad(2.0)
    res = c.icp_run_montecarlo(base, seed, 0, 1, n, ta, ra, "Ours", cfg, slots=256)
    recs = mc.records_from_results(np.arange(n), res)
    stats = mc.method_statistics(recs)
    assert stats["total_runs"] == n and 0.5 < stats["success_rate"] < 0.95
    tree = po.KdTree(pts)
    ocfg = po.default_config(search_radius=1.0, max_iterations=30, thresh_trans=1e-3, thresh_rot=1e-5, kappa_target=10.0, std_reg_gamma=100.0,
                             thres_cond=10.0, thres_eig=120.0, use_weight_derivative=1, always_compute_schur=1)
    for k in np.random.default_rng(7).choice(n, 16, replace=False):
        T0 = mc.trial_pose(base, int(k), seed, ta, ra)
        one, _ = c.icp_run(T0, "Ours", cfg)
        T = recs[k, mc.R_T:mc.R_T + 16].reshape(4, 4)
        assert recs[k, mc.R_ITERS] == one.iterations and recs[k, mc.R_CONV] == one.converged
        assert np.array_equal(T[:3, :3].reshape(9), np.array(one.R[:])) and np.array_equal(T[:3, 3], np.array(one.t[:]))
        ores, _ = po.icp_run(tree, pts, T0, "Ours", ocfg)
        assert ores.iterations == one.iterations and ores.converged == one.converged
        assert np.allclose(one.R[:], ores.R[:], atol=1e-7) and np.allclose(one.t[:], ores.t[:], atol=1e-7)
    c.close()



def test_the_order_of_the_source_only_schedules():
    """The source cloud is processed in the order of a space-filling curve (dcreg_set_source); stretching the curve's cells along x
    (debug option curve_x_scale) or keeping the caller's order changes which points share a wave and a partial row, i.e. the
    association of the 31 sums - nothing else: counts equal, sums to rounding, and each order agrees with the oracle."""
    tgt = h.scene_corridor(60_000, seed=21)
    src = (tgt[::3] + np.random.default_rng(22).normal(0, 0.01, tgt[::3].shape)).astype(np.float32)
    T = h.pose6d_matrix(0.05, -0.04, 0.02, h.deg2rad(0.2), h.deg2rad(-0.1), h.deg2rad(0.3))
    prm = api.default_lin_params(1.0, 1)
    tree = po.KdTree(tgt)
    ref = po.linearize(tree, src, T[:3, :3], T[:3, 3], po.default_lin_params(1.0, 1))
    outs = []
    for opt, val in (("curve_x_scale", 1.0), ("curve_x_scale", 0.0625), ("keep_source_order", 1.0)):
        c = api.Context(0)
        c.set_option(opt, val)
        c.set_target(tgt, 1.0); c.set_source(src)
        for Tk in (T, h.pose6d_matrix(0.04, -0.03, 0.02, h.deg2rad(0.15), h.deg2rad(-0.1), h.deg2rad(0.25)), T):      # cold, warm, back
            out = c.linearize(Tk[:3, :3], Tk[:3, 3], prm)
        outs.append(out)
        c.close()
        assert out["n_eff"] == ref["n_eff"] and out["n_pt"] == ref["n_pt"], (opt, val, out["n_eff"], ref["n_eff"])
        assert h.rel_err(out["H_upper"], ref["H_upper"]) < 1e-9 and h.rel_err(out["g"], ref["g"]) < 1e-8, (opt, val)
    for o in outs[1:]:
        assert h.rel_err(o["H_upper"], outs[0]["H_upper"]) < 1e-12
