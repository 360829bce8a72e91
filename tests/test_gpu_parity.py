"""GPU parity tests (run with -m gpu on an MI355X): the HIP path through the C-ABI vs the CPU oracle on
identical inputs, and vs the reference's committed traces.  Integer / index results are bit-exact; the
fp64 sums agree to rounding (tolerance 1e-9 relative, far inside north_star's 1e-5)."""
import os

import numpy as np
import pytest

import helpers as h
from dcreg_amd import api
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

H_RTOL = 1e-9   # relative to max|H| / max|g|


@pytest.fixture(scope="module")
def ctx():
    c = api.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def cyl():
    pts = h.cylinder_cloud()
    return pts, po.KdTree(pts)


def assert_lin_equal(gpu, ref):
    assert gpu["n_eff"] == ref["n_eff"] and gpu["n_pt"] == ref["n_pt"]
    assert h.rel_err(gpu["H_upper"], ref["H_upper"]) < H_RTOL
    assert h.rel_err(gpu["g"], ref["g"]) < 1e-8
    assert abs(gpu["sum_r2"] - ref["sum_r2"]) <= 1e-10 * max(1.0, ref["sum_r2"])
    assert abs(gpu["sum_b2"] - ref["sum_b2"]) <= 1e-10 * max(1.0, ref["sum_b2"])


def assert_debug_equal(gpu, ref):
    assert np.array_equal(gpu["flag"], ref["flag"])
    # neighbour lists are only defined (by the reference) for points whose 5th neighbour is inside the radius;
    # the GPU search stops at the radius, so compare those rows and all rows' flags
    ok = ref["flag"] != 0
    assert np.array_equal(gpu["nn_idx"][ok], ref["nn_idx"][ok])
    assert np.array_equal(gpu["nn_d2"][ok].view(np.uint32), ref["nn_d2"][ok].view(np.uint32))
    # per-point plane fits: same algorithm, different FMA contraction -> differences are rounding amplified by
    # the conditioning of the 5x3 LOAM system [q_j] x = -1 (planes tens of metres from the origin)
    passed = (ref["flag"] == 1) | (ref["flag"] == 4)
    assert np.allclose(gpu["normal"][passed], ref["normal"][passed], rtol=0, atol=1e-8)
    assert np.allclose(gpu["r"][passed], ref["r"][passed], rtol=0, atol=1e-8)
    assert np.allclose(gpu["s"][passed], ref["s"][passed], rtol=0, atol=1e-8)


@pytest.mark.parametrize("init,wd", [(h.RELEASE_INIT, 0), (h.PAPER_INIT, 1)])
def test_fixture_linearize_matches_oracle_and_golden(ctx, cyl, init, wd):
    pts, tree = cyl
    ctx.set_target(pts, 1.0)
    ctx.set_source(pts)
    T0 = h.pose6d_matrix(**init)
    gpu = ctx.linearize(T0[:3, :3], T0[:3, 3], api.default_lin_params(1.0, wd), debug=True)
    ref = po.linearize(tree, pts, T0[:3, :3], T0[:3, 3], po.default_lin_params(1.0, wd), debug=True)
    assert_lin_equal(gpu, ref)
    assert_debug_equal(gpu, ref)
    plain = ctx.linearize(T0[:3, :3], T0[:3, 3], api.default_lin_params(1.0, wd))
    assert np.array_equal(plain["H_upper"], gpu["H_upper"]) and np.array_equal(plain["g"], gpu["g"])
    if wd == 0:   # SURVEY Appendix B numbers (committed trace, iteration 0)
        assert gpu["n_eff"] == 871 and gpu["n_pt"] == 1557
        gold = [-47.16787056, 55.57558355, 4.97326544, 3.84171777, 4.98091287, -0.20608970]
        assert np.max(np.abs(-gpu["g"] - gold)) < 6e-9
    else:
        assert gpu["n_eff"] == 197


def test_deterministic_bitwise(ctx, cyl):
    pts, _ = cyl
    ctx.set_target(pts, 1.0)
    ctx.set_source(pts)
    T0 = h.pose6d_matrix(**h.PAPER_INIT)
    a = ctx.linearize(T0[:3, :3], T0[:3, 3], api.default_lin_params(1.0, 1))
    for _ in range(5):
        b = ctx.linearize(T0[:3, :3], T0[:3, 3], api.default_lin_params(1.0, 1))
        assert np.array_equal(a["H_upper"], b["H_upper"]) and np.array_equal(a["g"], b["g"])


SCENES = {
    "cylinder_100k": lambda: (h.scene_cylinder(100_000, seed=1, noise=0.01), 1.0),
    "planes_50k_r05": lambda: (h.scene_planes(50_000, seed=2), 0.5),
    "corridor_200k": lambda: (h.scene_corridor(200_000, seed=3), 1.0),
    "sparse_3k": lambda: (h.scene_cylinder(3_000, seed=4), 1.0),
}


@pytest.mark.parametrize("name", list(SCENES))
def test_synthetic_scene_parity(ctx, name):
    tgt, radius = SCENES[name]()
    rng = np.random.default_rng(11)
    src = tgt[rng.permutation(len(tgt))[: max(len(tgt) // 2, 1000)]].copy()
    src += rng.normal(0, 0.01, src.shape).astype(np.float32)
    T0 = h.pose6d_matrix(0.05, -0.08, 0.03, h.deg2rad(0.2), h.deg2rad(-0.1), h.deg2rad(0.5))
    ctx.set_target(tgt, radius)
    ctx.set_source(src)
    tree = po.KdTree(tgt)
    for wd in (0, 1):
        gpu = ctx.linearize(T0[:3, :3], T0[:3, 3], api.default_lin_params(radius, wd), debug=True)
        ref = po.linearize(tree, src, T0[:3, :3], T0[:3, 3], po.default_lin_params(radius, wd), debug=True)
        assert ref["n_eff"] > 10
        assert_lin_equal(gpu, ref)
        assert_debug_equal(gpu, ref)


def test_knn_exact_with_ties_and_outside_queries(ctx):
    """Lattice target (massive distance ties) + queries far outside the cloud: (d2, idx) order must match."""
    g = np.arange(0, 12, dtype=np.float32) * 0.25
    tgt = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    rng = np.random.default_rng(5)
    q = np.concatenate([tgt[:300] + 0.125, rng.uniform(-3, 6, (2000, 3)), [[100, 100, 100]], tgt[:50]]).astype(np.float32)
    ctx.set_target(tgt, 1.0)
    tree = po.KdTree(tgt)
    for k in (1, 5):
        gi, gd = ctx.knn(q, k=k, max_radius=0.0)
        oi, od = tree.knn(q, k=k)
        assert np.array_equal(gi, oi)
        assert np.array_equal(gd.view(np.uint32), od.view(np.uint32))
    # bounded search: everything inside the radius is exact, the rest is reported missing
    gi, gd = ctx.knn(q, k=5, max_radius=0.6)
    oi, od = tree.knn(q, k=5)
    inside = od <= np.float32(0.36)
    assert np.array_equal(gi[inside], oi[inside])
    assert np.all(gi[~inside] == -1)


@pytest.mark.parametrize("scene", ["lattice_ties", "cylinder_20k"])
def test_warm_start_bound_keeps_the_search_exact(ctx, scene):
    """The search of a linearisation is bounded by the previous call's neighbour sets (any pose).  Whatever the
    history - small steps, a jump far outside the cloud and back, a new target - every result must be bitwise the
    result of the unbounded (cold) search, including the (d2, idx) order among exact distance ties."""
    if scene == "lattice_ties":
        g = np.arange(0, 14, dtype=np.float32) * 0.25
        tgt = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
        src = (tgt[::3] + np.float32(0.125)).astype(np.float32)
    else:
        tgt = h.scene_cylinder(20_000, seed=7, noise=0.01)
        src = tgt[::2].copy()
    poses = [h.pose6d_matrix(0.05, -0.08, 0.03, 0.003, -0.002, 0.008), h.pose6d_matrix(0.03, -0.05, 0.02, 0.002, -0.001, 0.005),
             h.pose6d_matrix(0.0, 0.0, 0.0, 0.0, 0.0, 0.0), h.pose6d_matrix(500.0, 0.0, 0.0, 0.0, 0.0, 0.0),
             h.pose6d_matrix(0.3, 0.2, -0.1, 0.01, 0.02, -0.03), h.pose6d_matrix(0.0, 0.0, 0.0, 0.0, 0.0, 0.0)]
    prm = api.default_lin_params(1.0, 1)
    keys = ("H_upper", "g", "nn_idx", "flag", "r", "s", "normal")
    try:
        ctx.set_option("warm_start", 0)
        ctx.set_target(tgt, 1.0); ctx.set_source(src)
        cold = [ctx.linearize(T[:3, :3], T[:3, 3], prm, debug=True) for T in poses]
        ctx.set_option("warm_start", 1)
        ctx.set_target(tgt, 1.0); ctx.set_source(src)
        for k, T in enumerate(poses):
            w = ctx.linearize(T[:3, :3], T[:3, 3], prm, debug=(k % 2 == 0))    # debug and plain launches share the state
            assert w["n_eff"] == cold[k]["n_eff"] and w["n_pt"] == cold[k]["n_pt"]
            for key in keys:
                if key in w:
                    assert np.array_equal(w[key], cold[k][key]), (k, key)
            if "nn_d2" in w:
                assert np.array_equal(w["nn_d2"].view(np.uint32), cold[k]["nn_d2"].view(np.uint32))
        # a new target invalidates the stored positions (different sort order): results must follow the new cloud
        tgt2 = np.ascontiguousarray(tgt[::-1][: len(tgt) - 17])
        ctx.set_target(tgt2, 1.0)
        w = ctx.linearize(poses[0][:3, :3], poses[0][:3, 3], prm, debug=True)
        ref = po.linearize(po.KdTree(tgt2), src, poses[0][:3, :3], poses[0][:3, 3], po.default_lin_params(1.0, 1), debug=True)
        assert_lin_equal(w, ref)
        assert_debug_equal(w, ref)
    finally:
        ctx.set_option("warm_start", 1)


def test_empty_and_tiny_inputs(ctx):
    with pytest.raises(api.DcregError):
        ctx.set_target(np.zeros((0, 3), np.float32), 1.0)
    with pytest.raises(api.DcregError):
        ctx.set_source(np.zeros((0, 3), np.float32))
    tgt = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)   # fewer than 5 target points
    ctx.set_target(tgt, 1.0)
    ctx.set_source(tgt)
    out = ctx.linearize(np.eye(3), np.zeros(3))
    assert out["n_eff"] == 0 and out["n_pt"] == 0 and np.all(out["H_upper"] == 0)
    cfg = api.default_config()
    res, logs = ctx.icp_run(np.eye(4), "ME-SR", cfg)
    assert res.status == 1 and res.converged == 0 and res.iterations == 1 and logs == []


def test_batch_equals_single(ctx, cyl):
    pts, _ = cyl
    ctx.set_target(pts, 1.0)
    ctx.set_source(pts)
    rng = np.random.default_rng(0)
    Ts = [h.pose6d_matrix(*(rng.uniform(-0.3, 0.3, 3)), *(rng.uniform(-0.02, 0.02, 3))) for _ in range(7)]
    prm = api.default_lin_params(1.0, 1)
    outs = ctx.linearize_batch([T[:3, :3] for T in Ts], [T[:3, 3] for T in Ts], prm)
    for T, o in zip(Ts, outs):
        s = ctx.linearize(T[:3, :3], T[:3, 3], prm)
        assert s["n_eff"] == o["n_eff"] and np.array_equal(s["H_upper"], o["H_upper"]) and np.array_equal(s["g"], o["g"])


def _cfg(paper, **kw):
    base = dict(search_radius=1.0, max_iterations=30, CONVERGENCE_THRESH_TRANS=1e-3,
                CONVERGENCE_THRESH_ROT=1e-5 if paper else 1e-4, KAPPA_TARGET=10.0, STD_REG_GAMMA=100.0,
                use_weight_derivative=int(paper), always_compute_schur=int(paper))
    base.update(kw)
    return api.default_config(**base)


def _ocfg(paper, **kw):
    base = dict(search_radius=1.0, max_iterations=30, thresh_trans=1e-3, thresh_rot=1e-5 if paper else 1e-4,
                kappa_target=10.0, std_reg_gamma=100.0, use_weight_derivative=int(paper), always_compute_schur=int(paper))
    base.update(kw)
    return po.default_config(**base)


@pytest.mark.parametrize("family,method", [("release", m) for m in ("ME-SR", "ME-TSVD", "ME-TReg", "FCN-SR")] +
                         [("paper", m) for m in ("ME-SR", "ME-TSVD", "ME-TReg", "FCN-SR", "Ours")])
def test_engine_reproduces_committed_traces(ctx, cyl, family, method):
    """Full ICP runs through the HIP path vs the reference's committed per-iteration CSVs and the oracle."""
    pts, tree = cyl
    paper = family == "paper"
    init = h.PAPER_INIT if paper else h.RELEASE_INIT
    ctx.set_target(pts, 1.0)
    ctx.set_source(pts)
    res, logs = ctx.icp_run(h.pose6d_matrix(**init), method, _cfg(paper))
    ores, ologs = po.icp_run(tree, pts, h.pose6d_matrix(**init), method, _ocfg(paper))
    det = h.golden_rows(family, "iteration_details_with_dx.csv", method)
    cond = h.golden_rows(family, "condition_numbers_detailed.csv", method)
    allr = [r for r in h.golden_rows(family, "all_results.csv") if r["Method"] == method][0]
    assert res.iterations == ores.iterations == int(allr["Iterations"]) == len(det)
    assert res.converged == ores.converged == int(allr["Converged"])
    tol = 5e-7 if paper else 2e-7
    for L, O, d, c in zip(logs, ologs, det, cond):
        assert L.effective_points == O.n_eff == int(c["Effective_Points"])
        assert list(L.analysis.degenerate_mask[:]) == list(O.an.mask[:]) == [int(c["Degenerate_Mask_%d" % k]) for k in range(6)]
        assert np.allclose(L.update_dx[:], O.dx[:], rtol=1e-7, atol=1e-12)
        assert np.max(np.abs(np.array(L.update_dx[:]) - [float(d[k]) for k in ("dx_wx", "dx_wy", "dx_wz", "dx_x", "dx_y", "dx_z")])) < tol
        assert np.allclose(L.transform_matrix[:], O.T[:], rtol=0, atol=1e-11)
        assert h.rel_err(L.H_upper[:], O.H_upper[:]) < 1e-7
    assert np.isclose(logs[-1].trans_error_vs_gt, float(allr["Trans_Error_m"]), rtol=2e-5)
    assert np.isclose(logs[-1].rot_error_vs_gt, float(allr["Rot_Error_deg"]), rtol=2e-5)
    # final SE(3) pose vs the CPU path: north_star tolerance 1e-5 (we are ~1e-12)
    assert np.allclose(res.R[:], ores.R[:], atol=1e-10) and np.allclose(res.t[:], ores.t[:], atol=1e-10)


def test_batched_warm_states(ctx, cyl):
    """dcreg_reserve_warm_states / dcreg_linearize_batch_begin_warm: every pose of a batch keeps the warm-start state a single run
    keeps inside the ctx.  States only prune: results are bitwise those of the cold batch, whatever history the states hold;
    misuse (a state that was not reserved, one state for two poses of a launch) is refused."""
    pts, _ = cyl
    ctx.set_target(pts, 1.0)
    ctx.set_source(pts)
    rng = np.random.default_rng(7)
    prm = api.default_lin_params(1.0, 1)
    Ts = [h.pose6d_matrix(*(rng.uniform(-0.3, 0.3, 3)), *(rng.uniform(-0.02, 0.02, 3))) for _ in range(6)]
    Rs, ts = [T[:3, :3] for T in Ts], [T[:3, 3] for T in Ts]
    cold = ctx.linearize_batch(Rs, ts, prm)
    ctx.reserve_warm_states(8)
    for ids in ([0, 1, 2, 3, 4, 5], [5, 4, 3, 2, 1, 0], [7, -1, 6, -1, 0, 1]):      # fresh, scrambled (foreign histories), partly cold
        warm = ctx.linearize_batch_warm(Rs, ts, ids, prm)
        for a, b in zip(cold, warm):
            assert a["n_eff"] == b["n_eff"] and np.array_equal(a["H_upper"], b["H_upper"]) and np.array_equal(a["g"], b["g"])
    with pytest.raises(api.DcregError, match="not reserved"):
        ctx.linearize_batch_warm(Rs, ts, [0, 1, 2, 3, 4, 8], prm)
    with pytest.raises(api.DcregError, match="two poses"):
        ctx.linearize_batch_warm(Rs, ts, [0, 1, 2, 3, 1, 5], prm)
    ctx.set_source(pts)                                              # a new source drops the states
    with pytest.raises(api.DcregError, match="not reserved"):
        ctx.linearize_batch_warm(Rs, ts, [0, 1, 2, 3, 4, 5], prm)
    with pytest.raises(api.DcregError, match="65535"):
        ctx.linearize_batch([np.eye(3)] * 65536, [np.zeros(3)] * 65536, prm)


def test_trials_match_individual_runs(ctx, cyl):
    pts, _ = cyl
    ctx.set_target(pts, 1.0)
    ctx.set_source(pts)
    rng = np.random.default_rng(42)
    base = np.array([0.2, 0.8, 0.5, h.deg2rad(0.1), h.deg2rad(0.1), h.deg2rad(2.0)])
    T0s = []
    for _ in range(12):
        p = base + np.concatenate([rng.uniform(-0.3, 0.3, 3), rng.uniform(-1, 1, 3) * h.deg2rad(1.0)])
        T0s.append(h.pose6d_matrix(*p))
    cfg = _cfg(True)
    trials = ctx.icp_run_trials(T0s, "Ours", cfg)
    for T0, tr in zip(T0s, trials):
        res, logs = ctx.icp_run(T0, "Ours", cfg)
        assert tr.converged == res.converged and tr.iterations == res.iterations and tr.status == res.status
        T = np.array(tr.final_transform[:]).reshape(4, 4)
        assert np.array_equal(T[:3, :3].reshape(9), np.array(res.R[:])) and np.array_equal(T[:3, 3], np.array(res.t[:]))
        if logs:
            assert tr.corr_num == logs[-1].effective_points
            assert np.array_equal(np.array(tr.H_upper[:]), np.array(logs[-1].H_upper[:]))


def test_full_size_properties_1m(ctx):
    """BASELINE config 4 size (1M-point corridor): size-independent properties instead of a CPU replay --
    (i) H is PSD with the corridor-axis translation as its weakest direction, (ii) batch == single,
    (iii) splitting the source in two halves and adding the partial systems reproduces the whole
    (linearity of the reduction), (iv) counts are consistent."""
    tgt = h.scene_corridor(1_000_000, seed=9)
    rng = np.random.default_rng(1)
    src = tgt + rng.normal(0, 0.005, tgt.shape).astype(np.float32)
    T0 = h.pose6d_matrix(0.03, 0.02, -0.02, 0.0, 0.0, h.deg2rad(0.1))
    prm = api.default_lin_params(1.0, 0)
    ctx.set_target(tgt, 1.0)
    ctx.set_source(src)
    whole = ctx.linearize(T0[:3, :3], T0[:3, 3], prm)
    assert whole["n_pt"] == len(src) and 0.9 * len(src) < whole["n_eff"] <= whole["n_pt"]
    ev, V = np.linalg.eigh(whole["H"])
    assert ev[0] > 0 and abs(V[3, 0]) > 0.99          # weakest direction = translation along x
    assert ev[-1] / ev[0] > 50
    half = len(src) // 2
    parts = []
    for sl in (slice(0, half), slice(half, None)):
        ctx.set_source(src[sl])
        parts.append(ctx.linearize(T0[:3, :3], T0[:3, 3], prm))
    assert parts[0]["n_eff"] + parts[1]["n_eff"] == whole["n_eff"]
    assert h.rel_err(parts[0]["H_upper"] + parts[1]["H_upper"], whole["H_upper"]) < 1e-11
    assert h.rel_err(parts[0]["g"] + parts[1]["g"], whole["g"]) < 1e-9
    # spot-check 4000 random queries against the oracle's exact k-NN
    tree = po.KdTree(tgt)
    sel = rng.choice(len(src), 4000, replace=False)
    q = (src[sel].astype(np.float64) @ T0[:3, :3].T + T0[:3, 3]).astype(np.float32)
    gi, gd = ctx.knn(q, k=5, max_radius=1.0)
    oi, od = tree.knn(q, k=5)
    assert np.array_equal(gi, oi) and np.array_equal(gd.view(np.uint32), od.view(np.uint32))


def test_p2p_error_matches_oracle_and_golden(ctx, cyl):
    """dcreg_p2p_error (utils.hpp:538-589) vs the oracle and the committed all_results.csv (release run)."""
    pts, tree = cyl
    ctx.set_target(pts, 1.0)
    ctx.set_source(pts)
    for method in ("ME-SR", "ME-TReg"):
        res, logs = ctx.icp_run(h.pose6d_matrix(**h.RELEASE_INIT), method, _cfg(False))
        T = np.eye(4); T[:3, :3] = np.array(res.R[:]).reshape(3, 3); T[:3, 3] = res.t[:]
        rmse, fit, chamfer, valid = ctx.p2p_error(T, 0.2)
        aligned = (pts.astype(np.float64) @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
        ormse, ofit, ochamfer, ovalid = po.p2p_error(aligned, tree, 0.2)
        assert valid == ovalid
        assert np.isclose(rmse, ormse, rtol=1e-6) and np.isclose(fit, ofit, rtol=1e-12) and np.isclose(chamfer, ochamfer, rtol=1e-5)
        allr = [r for r in h.golden_rows("release", "all_results.csv") if r["Method"] == method][0]
        assert np.isclose(rmse, float(allr["P2P_RMSE"]), rtol=5e-5)
        assert np.isclose(fit, float(allr["P2P_Fitness"]), rtol=5e-5)
        assert np.isclose(chamfer, float(allr["Chamfer_Distance"]), rtol=5e-5)


def test_montecarlo_driver_on_gpu(ctx, cyl):
    """The Monte-Carlo driver fed by the batched GPU engine equals trial-by-trial oracle runs."""
    from dcreg_amd import montecarlo as mc
    pts, tree = cyl
    ctx.set_target(pts, 1.0)
    ctx.set_source(pts)
    cfg = _cfg(True, max_iterations=12)
    base = (0.2, 0.8, 0.5, h.deg2rad(0.1), h.deg2rad(0.1), h.deg2rad(2.0))
    recs, stats = mc.run_montecarlo(lambda T0s: ctx.icp_run_trials(T0s, "Ours", cfg), base, 9, seed=7, trans_amp=0.3,
                                    rot_amp_rad=h.deg2rad(1.0), batch=4)
    assert recs.shape == (9, mc.REC) and stats["total_runs"] == 9
    ocfg = _ocfg(True, max_iterations=12)
    for k in range(9):
        ores, ologs = po.icp_run(tree, pts, mc.trial_pose(base, k, 7, 0.3, h.deg2rad(1.0)), "Ours", ocfg)
        assert recs[k, mc.R_CONV] == ores.converged and recs[k, mc.R_ITERS] == ores.iterations
        T = recs[k, mc.R_T:mc.R_T + 16].reshape(4, 4)
        # north_star: final pose within 1e-5; the plane fits of the two paths differ by a few ulp, which the weakly constrained
        # fixture (lambda_min 0.6 against entries of 1e4) amplifies to ~1e-8 over a dozen iterations
        assert np.allclose(T[:3, :3].reshape(9), ores.R[:], atol=1e-7) and np.allclose(T[:3, 3], ores.t[:], atol=1e-7)
        if ologs:
            assert recs[k, mc.R_CORR] == ologs[-1].n_eff


def test_native_montecarlo_refills_slots_and_equals_single_runs(ctx, cyl):
    """dcreg_icp_run_montecarlo (poses generated in C++, `slots` trials in flight, a finished trial's slot - and its neighbour state,
    marked empty - taken over by the next trial at once): 41 trials through 6, 64 and 256 slots, as two interleaved rank shares, and
    through the Python-side driver give the same records bit for bit; each trial is bitwise the single run of its pose."""
    from dcreg_amd import montecarlo as mc
    pts, _ = cyl
    ctx.set_target(pts, 1.0)
    ctx.set_source(pts)
    cfg = _cfg(True, max_iterations=14)
    base = (0.2, 0.8, 0.5, h.deg2rad(0.1), h.deg2rad(0.1), h.deg2rad(2.0))
    n, seed, ta, ra = 41, 11, 0.4, h.deg2rad(1.5)
    ref, rstats = mc.run_montecarlo(lambda T0s: ctx.icp_run_trials(T0s, "Ours", cfg), base, n, seed=seed, trans_amp=ta, rot_amp_rad=ra, batch=256)
    cols = [c for c in range(mc.REC) if c != mc.R_TIME]
    for slots in (6, 64, 256):
        recs, stats = mc.run_montecarlo_native(ctx, "Ours", cfg, base, n, seed, ta, ra, slots=slots)
        assert np.array_equal(recs[:, cols], ref[:, cols]), slots
        assert {k: v for k, v in stats.items() if "time" not in k} == {k: v for k, v in rstats.items() if "time" not in k}
    halves = [mc.records_from_results(mc.shard_indices(n, r, 2), ctx.icp_run_montecarlo(base, seed, r, 2, len(mc.shard_indices(n, r, 2)), ta, ra, "Ours", cfg, slots=8))
              for r in (0, 1)]
    both = np.concatenate(halves)
    both = both[np.argsort(both[:, mc.R_TRIAL])]
    assert np.array_equal(both[:, cols], ref[:, cols])
    assert 0 < rstats["converged_runs"] and len(set(ref[:, mc.R_ITERS])) > 2          # trials of different lengths: slots did change hands
    for k in (0, 5, 17, 40):
        res, logs = ctx.icp_run(mc.trial_pose(base, k, seed, ta, ra), "Ours", cfg)
        T = ref[k, mc.R_T:mc.R_T + 16].reshape(4, 4)
        assert ref[k, mc.R_ITERS] == res.iterations and ref[k, mc.R_CONV] == res.converged
        assert np.array_equal(T[:3, :3].reshape(9), np.array(res.R[:])) and np.array_equal(T[:3, 3], np.array(res.t[:]))


def test_sharded_engine_plumbing(ctx, cyl):
    """dcreg_icp_run_sharded: with an identity reducer it IS dcreg_icp_run; with a reducer that adds a second, identical
    rank (row doubled, twice the source points) the unregularised update (2H)^-1 (2g) and the fitness are unchanged."""
    pts, _ = cyl
    ctx.set_target(pts, 1.0); ctx.set_source(pts)
    T0 = h.pose6d_matrix(**h.RELEASE_INIT)
    cfg = _cfg(False)
    ref, rlogs = ctx.icp_run(T0, "ME-SR", cfg)
    same, slogs = ctx.icp_run_sharded(T0, "ME-SR", cfg, len(pts), lambda row: None)
    assert same.iterations == ref.iterations and same.R[:] == ref.R[:] and same.t[:] == ref.t[:]
    assert all(a.update_dx[:] == b.update_dx[:] for a, b in zip(slogs, rlogs))
    calls = []

    def twice(row):
        calls.append(row[29])
        row *= 2.0
    one, ologs = ctx.icp_run(T0, "NONE", cfg)
    two, tlogs = ctx.icp_run_sharded(T0, "NONE", cfg, 2 * len(pts), twice)
    assert two.iterations == one.iterations == len(calls) and calls[0] == 871
    for a, b in zip(tlogs, ologs):
        assert a.effective_points == 2 * b.effective_points and a.fitness == b.fitness
        assert np.allclose(a.update_dx[:], b.update_dx[:], rtol=1e-9, atol=1e-15)

    def boom(row):
        raise ValueError("exchange failed")
    with pytest.raises(ValueError):
        ctx.icp_run_sharded(T0, "ME-SR", cfg, len(pts), boom)


def test_native_rccl_exchange_world_of_one(cyl):
    """dcreg_comm_* + dcreg_icp_run_sharded_rccl on the one GPU of the box: a communicator of one rank (the 8-GPU node runs the
    same code with world = 8): the all_gather returns the row, the engine run is bitwise dcreg_icp_run."""
    from dcreg_amd import pointshard as ps
    pts, _ = cyl
    c = api.Context(0)
    try:
        c.set_target(pts, 1.0); c.set_source(pts)
        with pytest.raises(api.DcregError):
            c.comm_allgather_sum(np.arange(32.0))                      # no communicator yet
        assert ps.init_native_exchange(c) == (0, 1)
        row = np.arange(32.0) * 0.5 - 3.0
        assert np.array_equal(c.comm_allgather_sum(row), row)
        T0 = h.pose6d_matrix(**h.PAPER_INIT)
        cfg = _cfg(True)
        a, la = c.icp_run(T0, "Ours", cfg)
        b, lb = c.icp_run_sharded_rccl(T0, "Ours", cfg, len(pts))
        assert a.iterations == b.iterations == 10 and a.converged == b.converged == 1
        assert np.array_equal(np.array(a.R[:]), np.array(b.R[:])) and np.array_equal(np.array(a.t[:]), np.array(b.t[:]))
        assert all(np.array_equal(np.array(x.H_upper[:]), np.array(y.H_upper[:])) and x.fitness == y.fitness for x, y in zip(la, lb))
        assert ps.init_native_exchange(c) == (0, 1)                      # re-initialisation replaces the communicator
    finally:
        c.close()


def test_sharded_run_never_leaves_a_rank_alone_in_the_exchange(cyl):
    """A rank whose linearisation fails (no target here) still takes part in the exchange with a poisoned row, and every rank
    stops together; an invalid total is refused before any exchange."""
    pts, _ = cyl
    c = api.Context(0)
    try:
        c.set_source(pts)                                             # target missing: dcreg_linearize would fail
        calls = []

        def reduce_rows(row):
            calls.append(row.copy())                                  # a second, healthy rank would add its row here
        res = api.IcpResult()
        with pytest.raises(api.DcregError):
            c.icp_run_sharded(h.pose6d_matrix(**h.PAPER_INIT), "Ours", _cfg(True), len(pts), reduce_rows)
        assert len(calls) == 1 and calls[0][31] == 1.0 and np.all(calls[0][:31] == 0.0)
        calls.clear()
        c.set_target(pts, 1.0)

        def poisoned_peer(row):
            calls.append(1)
            row[31] += 1.0                                            # another rank reports a failure
        with pytest.raises(api.DcregError):
            c.icp_run_sharded(h.pose6d_matrix(**h.PAPER_INIT), "Ours", _cfg(True), len(pts), poisoned_peer)
        assert calls == [1]                                           # stopped after the first exchange, not before it
        with pytest.raises(api.DcregError):
            c.icp_run_sharded(h.pose6d_matrix(**h.PAPER_INIT), "Ours", _cfg(True), 0, poisoned_peer)
        assert calls == [1]                                           # refused up front: no exchange entered
    finally:
        c.close()


def test_in_kernel_reduction_is_stable_under_uneven_load():
    """The fused reduction hands partial rows from block to block through agent-scope (sc1) stores / loads and a ticket, without
    release / acquire fences (a fence is a full L2 write-back on gfx950).  That protocol is one of the hand-off forms measured
    valid on this chip, but it is a property of the hardware, not of the language: hammer it.  6000 launches that reuse the same
    partial-row buffer, alternating poses (so a stale row would belong to the OTHER pose and change the sums), while a second
    context keeps the device unevenly busy with large launches; every result must be bitwise the first one of its pose."""
    import threading
    tgt = h.scene_cylinder(60_000, seed=21, noise=0.01)
    rng = np.random.default_rng(3)
    src = (tgt + rng.normal(0, 0.01, tgt.shape)).astype(np.float32)
    big = h.scene_corridor(400_000, seed=5)
    a, b = api.Context(0), api.Context(0)
    stop = threading.Event()
    try:
        a.set_target(tgt, 1.0); a.set_source(src)
        b.set_target(big, 1.0); b.set_source(big)

        def noise():
            k = 0
            while not stop.is_set():
                T = h.pose6d_matrix(0.01 * (k % 7), 0.0, 0.0, 0.0, 0.0, 0.001 * (k % 5))
                b.linearize(T[:3, :3], T[:3, 3], api.default_lin_params(1.0, 1))
                k += 1
        th = threading.Thread(target=noise)
        th.start()
        poses = [h.pose6d_matrix(0.05, -0.08, 0.03, 0.003, -0.002, 0.008), h.pose6d_matrix(-0.02, 0.04, 0.01, -0.001, 0.002, -0.004)]
        prm = api.default_lin_params(1.0, 1)
        out = api.LinOut()
        import ctypes as C
        Rs = [np.ascontiguousarray(T[:3, :3]).reshape(9) for T in poses]
        ts = [np.ascontiguousarray(T[:3, 3]) for T in poses]
        first = [None, None]
        for it in range(6000):
            k = it & 1
            assert a.linearize_raw(Rs[k], ts[k], prm, out) == 0
            row = (np.array(out.H_upper[:]), np.array(out.g[:]), out.sum_r2, out.sum_b2, out.n_eff, out.n_pt)
            if first[k] is None:
                first[k] = row
            else:
                assert np.array_equal(row[0], first[k][0]) and np.array_equal(row[1], first[k][1]) and row[2:] == first[k][2:], it
        assert not np.array_equal(first[0][0], first[1][0])
    finally:
        stop.set()
        th.join()
        a.close(); b.close()


def test_run_many_equals_individual_runs(cyl):
    """dcreg_icp_run_many: independent pairs on their own contexts / streams / host threads give exactly the results of
    running each pair alone."""
    pts, _ = cyl
    rng = np.random.default_rng(3)
    ctxs, T0s = [], []
    try:
        for q in range(3):
            c = api.Context(0)
            src = (pts + rng.normal(0, 0.002, pts.shape)).astype(np.float32)
            c.set_target(pts, 1.0); c.set_source(src)
            ctxs.append(c)
            T0s.append(h.pose6d_matrix(0.01 * (q + 1), 0.01, 0.01, 0.0, 0.0, 0.001 * q))
        cfg = _cfg(True)
        alone = [c.icp_run(T, "Ours", cfg)[0] for c, T in zip(ctxs, T0s)]
        many = api.icp_run_many(ctxs, np.stack(T0s), "Ours", cfg)
        for a, b in zip(alone, many):
            assert (a.converged, a.iterations, a.status) == (b.converged, b.iterations, b.status)
            assert a.R[:] == b.R[:] and a.t[:] == b.t[:]
        with pytest.raises(api.DcregError):
            api.icp_run_many([ctxs[0], ctxs[0]], np.stack(T0s[:2]), "Ours", cfg)     # a ctx is single-threaded
    finally:
        for c in ctxs:
            c.close()


def test_duplicate_points_and_rank_deficient_neighbourhoods(ctx):
    """Targets with exact duplicates: zero distances, distance ties resolved by index, and neighbour sets whose 5x3 system
    is rank deficient - cold and warm searches, all bit-exact against the oracle in indices, distances and gate flags."""
    base = h.scene_cylinder(24_000, seed=9, noise=0.0)
    tgt = np.concatenate([base, base[::7], base[::13], base[:200], base[:200], base[:200], base[:200]]).astype(np.float32)
    rng = np.random.default_rng(4)
    tgt = np.ascontiguousarray(tgt[rng.permutation(len(tgt))])
    src = np.concatenate([base[::4], base[:300] + np.float32(0.01)]).astype(np.float32)
    tree = po.KdTree(tgt)
    ctx.set_target(tgt, 1.0); ctx.set_source(src)
    for T in (h.pose6d_matrix(0.0, 0.0, 0.0, 0.0, 0.0, 0.0), h.pose6d_matrix(0.02, -0.01, 0.01, 0.001, 0.0, 0.002),
              h.pose6d_matrix(0.0, 0.0, 0.0, 0.0, 0.0, 0.0)):
        gpu = ctx.linearize(T[:3, :3], T[:3, 3], api.default_lin_params(1.0, 1), debug=True)
        ref = po.linearize(tree, src, T[:3, :3], T[:3, 3], po.default_lin_params(1.0, 1), debug=True)
        assert np.array_equal(gpu["flag"], ref["flag"])
        ok = ref["flag"] != 0
        assert np.array_equal(gpu["nn_idx"][ok], ref["nn_idx"][ok])
        assert np.array_equal(gpu["nn_d2"][ok].view(np.uint32), ref["nn_d2"][ok].view(np.uint32))
        assert gpu["n_eff"] == ref["n_eff"] and gpu["n_pt"] == ref["n_pt"]
        # plane fits: wherever the five neighbours are five distinct points the fits agree to rounding; neighbourhoods with
        # repeated points make the 5x3 system rank deficient, where the truncated-QR solution is decided by rounding noise in
        # the trailing pivots (in the reference's Eigen build as much as here), so only the gates are compared there
        passed = (ref["flag"] == 1) | (ref["flag"] == 4)
        nbr = tgt[np.clip(ref["nn_idx"], 0, None)]
        distinct = np.array([len({tuple(p) for p in row}) for row in nbr]) == 5
        sel = passed & distinct
        assert sel.sum() > 100 and (passed & ~distinct).sum() > 5
        assert np.allclose(gpu["normal"][sel], ref["normal"][sel], rtol=0, atol=1e-7)
        assert np.allclose(gpu["r"][sel], ref["r"][sel], rtol=0, atol=1e-7)


def test_empty_space_skip_is_exact(ctx):
    """Queries in the empty space between two clusters: the distance field lets them skip the rings it knows are empty;
    results must equal the plain ring walk and the oracle (bounded and unbounded searches)."""
    rng = np.random.default_rng(8)
    a = rng.uniform(0, 2, (4000, 3)); b = rng.uniform(0, 2, (4000, 3)) + np.array([9.0, 0.5, 0.0])
    tgt = np.concatenate([a, b]).astype(np.float32)
    q = np.concatenate([rng.uniform(2, 9, (1500, 3)) * np.array([1, 0.3, 0.3]), rng.uniform(-4, 14, (500, 3)), tgt[:100] + 0.01]).astype(np.float32)
    tree = po.KdTree(tgt)
    oi, od = tree.knn(q, k=5)
    got = {}
    try:
        for opt in (1, 0):
            ctx.set_option("gap_field", opt)
            ctx.set_target(tgt, 3.0)                      # radius 3 m: up to a dozen rings of ~0.25 m cells
            got[opt] = (ctx.knn(q, k=5, max_radius=0.0), ctx.knn(q, k=5, max_radius=3.0), ctx.index_info().cell)
    finally:
        ctx.set_option("gap_field", 1)
    for opt in (1, 0):
        (gi, gd), (bi, bd), cell = got[opt]
        assert cell < 1.0
        assert np.array_equal(gi, oi) and np.array_equal(gd.view(np.uint32), od.view(np.uint32))
        inside = od < np.float32(9.0)
        assert np.array_equal(bi[inside], oi[inside])
    assert np.array_equal(got[0][1][0], got[1][1][0])


def test_certificates_only_spare_searches(cyl):
    """Certificates decide, per point and launch, whether the 6-NN search can be skipped.  Either way the sums are bitwise the same:
    a walk that mixes tiny steps, jumps and a trip far outside the cloud, then two pipelined ICP runs, with certificates used and not
    used ("use_certificates" 0: every point is searched in every launch) agree bit for bit - and with them most points are left
    unsearched once the pose has settled."""
    tgt = cyl[0]
    src = tgt[::2]
    T0 = h.pose6d_matrix(0.3, -0.2, 0.1, h.deg2rad(2.0), h.deg2rad(-1.0), h.deg2rad(3.0))
    cfg = api.default_config(search_radius=1.0, max_iterations=25, CONVERGENCE_THRESH_ROT=0.0, CONVERGENCE_THRESH_TRANS=0.0)
    prm = api.default_lin_params(1.0, 1)
    walk = [T0, T0 @ h.pose6d_matrix(1e-5, 0.0, 2e-5, 1e-7, 0.0, -1e-7), T0 @ h.pose6d_matrix(0.5, 0.5, -0.2, 0.0, h.deg2rad(4.0), 0.0), T0,
            T0 @ h.pose6d_matrix(300.0, 0.0, 0.0, 0.0, 0.0, 0.0), T0, T0]
    got = {}
    for use in (0, 1):
        c = api.Context(0)
        c.set_option("use_certificates", use); c.set_option("count_searches", 1)
        c.set_target(tgt, 1.0); c.set_source(src)
        lin, searched = [], []
        for T in walk:
            lin.append(c.linearize(T[:3, :3], T[:3, 3], prm))
            searched.append(c.launch_stats(reset=True)["points_searched"])
        res, logs = c.icp_run(T0, "Ours", cfg)                      # pipelined engine, state carried over from the walk
        st1 = c.launch_stats(reset=True)
        res2, _ = c.icp_run(T0, "Ours", cfg)                        # and once more from the converged state
        st2 = c.launch_stats(reset=True)
        got[use] = (lin, res.iterations, np.array(res.R[:]), np.array(res.t[:]), np.array(res2.R[:]), np.array(res2.t[:]),
                    [np.array(L.H_upper[:]) for L in logs], searched, st1, st2)
        c.close()
    a, b = got[1], got[0]
    for x, y in zip(a[0], b[0]):
        assert x["n_eff"] == y["n_eff"] and np.array_equal(x["H_upper"], y["H_upper"]) and np.array_equal(x["g"], y["g"])
    assert a[1] == b[1] and all(np.array_equal(a[k], b[k]) for k in (2, 3, 4, 5))
    assert all(np.array_equal(x, y) for x, y in zip(a[6], b[6]))
    n = len(src)
    assert b[7] == [n] * len(walk) and b[8]["points_searched"] == b[8]["points"] == 25 * n      # without: everything, always
    assert a[7][0] == n and a[7][1] < 0.01 * n and a[7][6] == 0                                  # fresh / a 20 um step / the same pose again
    assert a[7][4] == n                                                                          # 300 m away: every certificate fails
    assert a[8]["points_searched"] < 0.75 * a[8]["points"] and a[9]["points_searched"] < 0.75 * a[9]["points"]      # (25 iterations from 0.4 m / 3 deg off)


def test_kdtree_comparator_returns_what_the_grid_returns():
    """The kd-tree over the target (csrc/device/kdtree.hip, a comparator of the grid index: dcreg_debug.h) answers exact k-NN queries
    with the same lists as the grid, bit for bit: aligned and misaligned queries, queries far outside the cloud, with and without a
    radius, k = 1 and 5, a lattice with duplicated points (ties: the (d2, index) order decides), several leaf sizes."""
    rng = np.random.default_rng(31)
    g = np.arange(0, 12, dtype=np.float32) * 0.25
    lattice = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    cases = {"corridor": (h.scene_corridor(120_000, seed=3, length=40.0), 1.0),
             "fixture": (h.cylinder_cloud(), 1.0),
             "lattice_dups": (np.concatenate([lattice, lattice[::5]]), 0.6)}
    for name, (tgt, radius) in cases.items():
        c = api.Context(0)
        c.set_target(tgt, radius)
        T = h.pose6d_matrix(0.4, -0.5, 0.3, h.deg2rad(1.0), h.deg2rad(-2.0), h.deg2rad(3.0))
        base = tgt[rng.permutation(len(tgt))[:20_000]]
        queries = [base + rng.normal(0, 0.01, base.shape).astype(np.float32),
                   (base.astype(np.float64) @ T[:3, :3].T + T[:3, 3]).astype(np.float32),
                   base + np.float32(300.0), base if name == "lattice_dups" else base[:1000] * np.float32(0.5)]
        for leaf in (4, 16, 64):
            depth, _, _ = c.kdtree_build(leaf)
            assert (leaf << depth) >= len(tgt)
            for q in queries:
                for k in (1, 5):
                    for r in (0.0, radius):
                        ig, dg, _ = c.knn_timed(q, k, r, "grid", repeats=1)
                        ik, dk, _ = c.knn_timed(q, k, r, "kdtree", repeats=1)
                        assert np.array_equal(ig, ik) and np.array_equal(dg.view(np.uint32), dk.view(np.uint32)), (name, leaf, k, r)
                        if k == 5 and r > 0.0 and leaf == 16:       # the row sweep of the linearisation in the plain kernel
                            i2, d2_, _ = c.knn_timed(q, k, r, "grid_sweep", repeats=1)
                            assert np.array_equal(ig, i2) and np.array_equal(dg.view(np.uint32), d2_.view(np.uint32)), (name, "sweep")
        i0, d0 = c.knn(queries[1], 5, radius)
        i1, d1, _ = c.knn_timed(queries[1], 5, radius, "kdtree", repeats=1)
        assert np.array_equal(i0, i1) and np.array_equal(d0.view(np.uint32), d1.view(np.uint32))
        c.close()


def test_dispatch_order_only_schedules():
    """Launches with more query blocks than the device holds at once hand them out heaviest group first (kernels.hpp k_group_cost: an
    estimate per cloud pair at the pose of the first launch, used while the misalignment hint says it matters).  Scheduling only: with
    the order on, off, estimated at another pose, with hints that switch it on and off between launches, every launch and a pipelined
    ICP run give bitwise the same sums."""
    tgt = h.scene_corridor(400_000, seed=21, length=80.0)
    rng = np.random.default_rng(22)
    src = (tgt + rng.normal(0, 0.01, tgt.shape)).astype(np.float32)
    T0 = h.pose6d_matrix(0.05, -0.08, 0.03, h.deg2rad(0.2), h.deg2rad(-0.1), h.deg2rad(1.0))        # 0.7 m at the ends
    walk = [T0, T0 @ h.pose6d_matrix(0.01, 0.0, 0.0, 0.0, 0.0, 1e-4), np.eye(4), T0]
    cfg = api.default_config(search_radius=1.0, max_iterations=12, CONVERGENCE_THRESH_ROT=0.0, CONVERGENCE_THRESH_TRANS=0.0)
    prm = api.default_lin_params(1.0, 1)
    got = {}
    for mode in ("off", "on", "other_pose", "hints"):
        c = api.Context(0)
        c.set_option("dispatch_order", 0 if mode == "off" else 1)
        c.set_target(tgt, 1.0); c.set_source(src)
        if mode == "other_pose":                                    # the estimate is made here, 3 m off: a poor order, still only an order
            Tx = h.pose6d_matrix(3.0, 0.5, 0.0, 0.0, 0.0, h.deg2rad(-2.0))
            c.linearize(Tx[:3, :3], Tx[:3, 3], prm)
        lin = []
        for k, T in enumerate(walk):
            if mode == "hints":
                c.hint_misalignment([10.0, 0.0, 1e-3, float("nan")][k])
            lin.append(c.linearize(T[:3, :3], T[:3, 3], prm))
        res, logs = c.icp_run(T0, "Ours", cfg)
        got[mode] = (lin, np.array(res.R[:]), np.array(res.t[:]), [np.array(L.H_upper[:]) for L in logs])
        c.close()
    ref = got["off"]
    for mode in ("on", "other_pose", "hints"):
        x = got[mode]
        for a, b in zip(x[0], ref[0]):
            assert a["n_eff"] == b["n_eff"] and np.array_equal(a["H_upper"], b["H_upper"]) and np.array_equal(a["g"], b["g"]), mode
        assert np.array_equal(x[1], ref[1]) and np.array_equal(x[2], ref[2]) and all(np.array_equal(p, q) for p, q in zip(x[3], ref[3])), mode


def test_a_wait_that_runs_out_of_patience_is_harmless(cyl):
    """"wait_seconds": when a result has not arrived after that long, the waiting host calls off the launch queued behind it (a gate
    that would otherwise wait for this very thread), drains the stream to surface a device fault, and carries on when the result is
    there after all (a slow but healthy launch).  With no patience at all every wait of the pipelined engine takes that path: the run
    must come out bitwise as usual, and the context must stay usable."""
    tgt = cyl[0]
    src = tgt[::2]
    T0 = h.pose6d_matrix(0.3, -0.2, 0.1, h.deg2rad(2.0), h.deg2rad(-1.0), h.deg2rad(3.0))
    cfg = api.default_config(search_radius=1.0, max_iterations=12, CONVERGENCE_THRESH_ROT=0.0, CONVERGENCE_THRESH_TRANS=0.0)
    got = []
    for patience in (30.0, 1e-9):
        c = api.Context(0)
        c.set_option("wait_seconds", patience); c.set_target(tgt, 1.0); c.set_source(src)
        res, logs = c.icp_run(T0, "Ours", cfg)
        res2, _ = c.icp_run(T0, "Ours", cfg)
        lin = c.linearize(T0[:3, :3], T0[:3, 3], api.default_lin_params(1.0, 1))
        got.append((res.iterations, np.array(res.R[:]), np.array(res.t[:]), np.array(res2.R[:]), [np.array(L.H_upper[:]) for L in logs], lin["H_upper"]))
        c.close()
    a, b = got
    assert a[0] == b[0] == 12 and all(np.array_equal(a[k], b[k]) for k in (1, 2, 3, 5))
    assert all(np.array_equal(x, y) for x, y in zip(a[4], b[4]))


def test_far_from_the_origin_and_very_dense_cells(ctx):
    """Coordinates ~1e5 m from the origin (float spacing 8 mm: heavy quantisation, exact ties, cell arithmetic in double) and a cloud
    whose 60 k points sit in a 2 cm cube (runs of tens of thousands of points per cell): exact k-NN and a linearisation vs the oracle."""
    rng = np.random.default_rng(21)
    base = h.scene_corridor(20000, seed=3)[:, :3]
    far = (base.astype(np.float64) + np.array([1.0e5, -2.0e5, 3.0e4])).astype(np.float32)
    q = np.concatenate([far[::7] + rng.normal(0, 0.05, far[::7].shape).astype(np.float32), far[:200]]).astype(np.float32)
    tree = po.KdTree(far)
    oi, od = tree.knn(q, k=5)
    ctx.set_target(far, 1.0)
    gi, gd = ctx.knn(q, k=5, max_radius=0.0)
    assert np.array_equal(gi, oi) and np.array_equal(gd.view(np.uint32), od.view(np.uint32))
    src = far[::3]
    ctx.set_source(src)
    T = h.pose6d_matrix(0.02, -0.01, 0.01, h.deg2rad(1e-5), 0.0, h.deg2rad(-1e-5))      # tiny: the lever arm is 2e5 m
    gpu = ctx.linearize(T[:3, :3], T[:3, 3], api.default_lin_params(1.0, 1), debug=True)
    ref = po.linearize(tree, src, T[:3, :3], T[:3, 3], po.default_lin_params(1.0, 1), debug=True)
    assert gpu["n_eff"] == ref["n_eff"] and gpu["n_pt"] == ref["n_pt"] and np.array_equal(gpu["flag"], ref["flag"])
    ok = ref["flag"] != 0
    assert np.array_equal(gpu["nn_idx"][ok], ref["nn_idx"][ok])
    dense = (rng.uniform(0, 0.02, (60000, 3)) + np.array([5.0, 5.0, 5.0])).astype(np.float32)
    qd = np.concatenate([dense[::300] + 0.001, rng.uniform(4.5, 5.5, (300, 3)), [[5.0, 5.0, 5.0]]]).astype(np.float32)
    oi, od = po.KdTree(dense).knn(qd, k=5)
    ctx.set_target(dense, 1.0)
    gi, gd = ctx.knn(qd, k=5, max_radius=0.0)
    assert np.array_equal(gi, oi) and np.array_equal(gd.view(np.uint32), od.view(np.uint32))
    bi, bd = ctx.knn(qd, k=5, max_radius=0.3)
    inside = od < np.float32(0.09)
    assert np.array_equal(bi[inside], oi[inside])


def test_gated_launches_equal_blocking_ones(cyl):
    """dcreg_linearize_gated_begin / _gate_open / _gate_abort (the launch pipeline of dcreg_icp_run): a linearisation queued behind
    the gate before its pose exists gives bitwise the result of the blocking call; a launch that is called off leaves results and
    warm state untouched; the protocol errors are reported, and a context can be destroyed with a gate still waiting."""
    tgt = cyl[0]
    src = tgt[::3]
    prm = api.default_lin_params(1.0, 1)
    poses = [h.pose6d_matrix(0.05 * k, -0.03 * k, 0.02, h.deg2rad(0.3 * k), 0.0, h.deg2rad(-0.2 * k)) for k in range(6)]
    ref = api.Context(0)
    ref.set_target(tgt, 1.0); ref.set_source(src)
    want = [ref.linearize(T[:3, :3], T[:3, 3], prm) for T in poses]
    c = api.Context(0)
    c.set_target(tgt, 1.0); c.set_source(src)
    slot = 0
    c.linearize_begin(poses[0][:3, :3], poses[0][:3, 3], prm, slot=slot)
    got = []
    for k in range(len(poses)):
        last = k + 1 == len(poses)
        if not last:
            c.linearize_gated_begin(prm, slot=slot ^ 1)             # queued while linearisation k is in flight
            with pytest.raises(api.DcregError):
                c.linearize_gated_begin(prm, slot=slot)             # only one gate at a time
            with pytest.raises(api.DcregError):
                c.linearize(poses[0][:3, :3], poses[0][:3, 3], prm) # nothing else may queue behind a waiting gate
        got.append(c.linearize_end(slot=slot))
        if not last:
            c.gate_open(poses[k + 1][:3, :3], poses[k + 1][:3, 3])
            slot ^= 1
    for a, b in zip(got, want):
        assert a["n_eff"] == b["n_eff"] and np.array_equal(a["H_upper"], b["H_upper"]) and np.array_equal(a["g"], b["g"])
    with pytest.raises(api.DcregError):
        c.gate_open(poses[0][:3, :3], poses[0][:3, 3])              # nothing waits
    # a launch that is called off: the next blocking call behaves as if it had never been queued
    c.linearize_gated_begin(prm, slot=1)
    c.gate_abort()
    c.gate_abort()                                                  # idempotent
    again = c.linearize(poses[-1][:3, :3], poses[-1][:3, 3], prm)
    assert again["n_eff"] == want[-1]["n_eff"] and np.array_equal(again["H_upper"], want[-1]["H_upper"])
    # called off, and the next gated launch published at once: the first gate may only start after the record already carries the
    # later number - it must read that as its own abort, not wait for a number that will never come
    for _ in range(20):
        c.linearize_gated_begin(prm, slot=1)
        c.gate_abort()
        c.linearize_gated_begin(prm, slot=1)
        c.gate_open(poses[-1][:3, :3], poses[-1][:3, 3])
        out = c.linearize_end(slot=1)
        assert out["n_eff"] == want[-1]["n_eff"] and np.array_equal(out["H_upper"], want[-1]["H_upper"])
    c.linearize_gated_begin(prm, slot=1)
    c.close()                                                       # destroys the context with the gate still waiting
    ref.close()
    # without the pinned-flag wait ("spin" 0) a stream synchronise would wait for the gate: the launch is refused and the engine
    # falls back to blocking launches - same run, bit for bit
    T0 = h.pose6d_matrix(0.3, -0.2, 0.1, h.deg2rad(2.0), h.deg2rad(-1.0), h.deg2rad(3.0))
    cfg = api.default_config(search_radius=1.0, max_iterations=12)
    runs = []
    for spin in (1, 0):
        e = api.Context(0)
        e.set_option("spin", spin); e.set_target(tgt, 1.0); e.set_source(src)
        if not spin:
            with pytest.raises(api.DcregError):
                e.linearize_gated_begin(prm, slot=1)
        res, logs = e.icp_run(T0, "Ours", cfg)
        runs.append((res.iterations, np.array(res.R[:]), np.array(res.t[:])))
        e.close()
    assert runs[0][0] == runs[1][0] and np.array_equal(runs[0][1], runs[1][1]) and np.array_equal(runs[0][2], runs[1][2])


def test_x_sub_cells_only_trim_the_candidate_runs(ctx, cyl):
    """The "x_subdiv" option cuts every grid cell into sub-cells along x (finer trimming of the candidate runs, same rows):
    k-NN results must stay exact for every value, and a linearisation bitwise the same (same neighbour sets, same sum order)."""
    g = np.arange(0, 12, dtype=np.float32) * 0.25
    tgt = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    tgt = np.concatenate([tgt, tgt[::9]])
    rng = np.random.default_rng(6)
    q = np.concatenate([tgt[:300] + 0.125, rng.uniform(-3, 6, (2000, 3)), [[100, 100, 100]], [[-250, 3, 1]]]).astype(np.float32)
    oi, od = po.KdTree(tgt).knn(q, k=5)
    tgt_c = cyl[0]
    src_c = tgt_c[::2]
    T0 = h.pose6d_matrix(0.3, -0.2, 0.1, h.deg2rad(2.0), h.deg2rad(-1.0), h.deg2rad(3.0))
    sums = {}
    try:
        for sx in (1, 2, 4, 8, 16):
            ctx.set_option("x_subdiv", sx)
            ctx.set_target(tgt, 1.0)
            gi, gd = ctx.knn(q, k=5, max_radius=0.0)
            assert np.array_equal(gi, oi) and np.array_equal(gd.view(np.uint32), od.view(np.uint32)), sx
            ctx.set_target(tgt_c, 1.0); ctx.set_source(src_c)
            out = ctx.linearize(T0[:3, :3], T0[:3, 3], api.default_lin_params(1.0, 1))
            sums[sx] = (out["n_eff"], np.array(out["H_upper"]), np.array(out["g"]))
    finally:
        ctx.set_option("x_subdiv", 8)
    for sx in (2, 4, 8, 16):
        assert sums[sx][0] == sums[1][0] and np.array_equal(sums[sx][1], sums[1][1]) and np.array_equal(sums[sx][2], sums[1][2]), sx


@pytest.mark.parametrize("method,n_iter", [("Ours", 1500), ("ME-SR", 300), ("ME-TSVD", 300), ("ME-TReg", 300), ("FCN-SR", 300)])
def test_fig8_long_trace_through_the_hip_path(ctx, cyl, method, n_iter):
    """icp_iter.yaml (max_iterations 5000, vanishing thresholds): the committed per-iteration history of the reference,
    reproduced through dcreg_icp_run - correspondence count exact on every one of the 1500 / 300 iterations (each is a
    warm-started search whose bound comes from the iteration before), errors and rmse to the printed precision."""
    import os
    pts, _ = cyl
    ctx.set_target(pts, 1.0); ctx.set_source(pts)
    cfg = _cfg(True, max_iterations=n_iter, CONVERGENCE_THRESH_TRANS=1e-12, CONVERGENCE_THRESH_ROT=1e-14)
    res, logs = ctx.icp_run(h.pose6d_matrix(**h.PAPER_INIT), method, cfg)
    rows = [r for r in h.read_csv_rows(os.path.join(h.GOLDEN, "fig8", "iteration_history.csv.gz")) if r["Method"] == method][:n_iter]
    assert len(logs) == n_iter == len(rows)
    mism = 0
    for L, r in zip(logs, rows):
        mism += int(L.effective_points != int(r["CorrNum"]))
        assert abs(L.trans_error_vs_gt - float(r["TransError"])) < 2e-6
        assert abs(L.rot_error_vs_gt - float(r["RotError"])) < 2e-5
        assert abs(L.rmse - float(r["RMSE"])) < 2e-6
    assert mism == 0


def test_montecarlo_cli_runs_and_is_seeded():
    from dcreg_amd import montecarlo as mc
    a = mc.main(["--trials", "48", "--methods", "Ours,ME-SR", "--batch", "32"])
    b = mc.main(["--trials", "48", "--methods", "Ours", "--batch", "17"])       # batching must not change a trial
    assert a["Ours"]["total_runs"] == 48 and 0.0 <= a["ME-SR"]["success_rate"] <= 1.0
    for k in ("success_rate", "mean_trans_error", "mean_rot_error", "mean_iterations"):
        assert a["Ours"][k] == b["Ours"][k]
