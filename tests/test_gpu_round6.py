"""GPU tests added in round 6 (run with -m gpu on an MI355X; everything through the C-ABI):
  * clouds with non-finite coordinates are refused, source and target alike (NaN as well as infinity);
  * dcreg_reset_warm_state(ctx, -1) drops the context's own neighbour state: the next run is bitwise the run of a fresh context (the
    reference's fresh ICPContext per run, icp_test_runner.cpp:408-409 - what bench.py's `cold_run` times);
  * small frames from host buffers go through the context's pinned block: the caller's buffer may be overwritten as soon as
    dcreg_set_source returns;
  * the linearisation kernel in one-wave blocks, the Monte-Carlo job, the 5 M-point prior map against the oracle;
  * the WINDOW index of a large map (context.hpp): bitwise invisible - walks in and out of the window, whole pipelined runs incl. one whose
    every pose leaves the window, k-NN / metrics / batches on the whole map - and engaged by itself when the table budget binds."""
import numpy as np
import pytest

import helpers as h
from dcreg_amd import api

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bad", [np.nan, np.inf, -np.inf])
@pytest.mark.parametrize("n", [5_000, 120_000])          # the host-bounds path of small frames and the device-bounds path
def test_non_finite_clouds_are_refused(bad, n):
    tgt = h.scene_cylinder(n, seed=3, noise=0.01)
    ctx = api.Context(0)
    try:
        ctx.set_target(tgt, 1.0)
        ctx.set_source(tgt[::2].copy())
        good = ctx.linearize(np.eye(3), np.zeros(3), api.default_lin_params(1.0, 0))
        for axis in range(3):
            src = tgt[::2].copy()
            src[len(src) // 3, axis] = bad
            with pytest.raises(api.DcregError) as e:
                ctx.set_source(src)
            assert "non-finite" in str(e.value)
            t2 = tgt.copy()
            t2[len(t2) // 5, axis] = bad
            with pytest.raises(api.DcregError) as e:
                ctx.set_target(t2, 1.0)
            assert "non-finite" in str(e.value)
        # the context is still usable afterwards
        ctx.set_target(tgt, 1.0)
        ctx.set_source(tgt[::2].copy())
        again = ctx.linearize(np.eye(3), np.zeros(3), api.default_lin_params(1.0, 0))
        assert again["n_eff"] == good["n_eff"] and np.array_equal(again["H_upper"], good["H_upper"])
    finally:
        ctx.close()


def test_dropping_the_own_state_gives_the_run_of_a_fresh_context():
    tgt = h.scene_corridor(200_000, seed=9)
    rng = np.random.default_rng(1)
    src = (tgt + rng.normal(0, 0.01, tgt.shape)).astype(np.float32)
    T0 = h.pose6d_matrix(0.05, -0.08, 0.03, h.deg2rad(0.2), h.deg2rad(-0.1), h.deg2rad(0.5))
    cfg = api.default_config(search_radius=1.0, max_iterations=30, KAPPA_TARGET=10.0, STD_REG_GAMMA=100.0, use_weight_derivative=1, always_compute_schur=1)

    def run(ctx):
        res, logs = ctx.icp_run(T0, "Ours", cfg)
        return res, [(L.effective_points, L.corr_pt_count, tuple(L.update_dx[:]), tuple(L.H_upper[:])) for L in logs]

    fresh = api.Context(0)
    fresh.set_option("record_launches", 1)
    fresh.set_target(tgt, 1.0); fresh.set_source(src)
    r0, l0 = run(fresh)
    s0 = fresh.launch_series(reset=True)
    r1, l1 = run(fresh)                 # warm: starts from what the converged pose left
    s1 = fresh.launch_series(reset=True)
    fresh.reset_warm_state(-1)
    r2, l2 = run(fresh)                 # cold again
    s2 = fresh.launch_series(reset=True)
    fresh.close()
    assert r0.converged == 1 and r0.iterations == r1.iterations == r2.iterations
    assert l0 == l1 == l2                                               # history independence: bitwise the same run every time
    assert s0["searched"][0] == s2["searched"][0] == len(src)           # a cold first launch searches every point ...
    assert np.array_equal(s0["searched"], s2["searched"])               # ... and the dropped state behaves like the fresh one launch by launch
    c2 = api.Context(0)
    with pytest.raises(api.DcregError):
        c2.reset_warm_state(0)                                          # (no batch states reserved: only -1 names a state)
    c2.close()


def test_small_host_frames_are_consumed_when_set_source_returns():
    tgt = h.scene_cylinder(60_000, seed=2, noise=0.01)
    frame = tgt[::8].copy()
    prm = api.default_lin_params(1.0, 0)
    ctx = api.Context(0)
    try:
        ctx.set_target(tgt, 1.0)
        ctx.set_source(frame)
        want = ctx.linearize(np.eye(3), np.zeros(3), prm)
        for stride_pad in (0, 1):        # xyz and xyzi layouts
            buf = np.zeros((len(frame), 3 + stride_pad), np.float32)
            for rep in range(20):
                buf[:, :3] = frame
                ctx.set_source(buf)                               # (float32, contiguous: passed as is, stride 3 or 4)
                buf[:] = 1e9                                      # scribble over the caller's buffer at once
                got = ctx.linearize(np.eye(3), np.zeros(3), prm)
                assert got["n_eff"] == want["n_eff"] and np.array_equal(got["H_upper"], want["H_upper"]) and np.array_equal(got["g"], want["g"])
    finally:
        ctx.close()


# ---------------------------------------------------------------- the reference's published regime: a small frame against a large prior map
from oracle import pyoracle as po                                                          # noqa: E402  (the checker)
from test_gpu_parity import assert_lin_equal, assert_debug_equal                           # noqa: E402
from test_gpu_configs import assert_runs_equal, assert_cov_equal, cfg_pair                 # noqa: E402


@pytest.mark.timeout(1200)
def test_registration_against_a_5m_point_prior_map_matches_the_oracle():
    """8 k-point frame against a 5 M-point seeded prior map (scenes.scene_prior_map: the largest map the oracle's kd-tree handles inside the
    test timeout; bench.py's c3_prior_map_50m runs the 50 M-point one), R = 0.5, Ours, the yaml's poses and thresholds: neighbour lists, float
    distances and gate flags of the first linearisation bit for bit; the whole run iteration by iteration; with the dense cell table capped at
    2^22 entries (the cell edge grows, x sub-cells go: rounds 1-5's behaviour at scale) every iteration's sums are bitwise the same."""
    tgt, src = h.scene_prior_map(5_000_000, extent=110.0)
    gt, T0 = h.pose6d_matrix(**h.PK01_GT), h.pose6d_matrix(**h.PK01_INIT)
    tree = po.KdTree(tgt)
    cfg, ocfg = cfg_pair(0.5, 30, 0, 1e-5, 1e-3, gt.reshape(16))
    runs = {}
    for name, entries in (("dense", 0), ("capped", 1 << 22)):
        ctx = api.Context(0)
        try:
            if entries:
                ctx.set_option("max_table_entries", entries)
            ctx.set_target(tgt, 0.5)
            ctx.set_source(src)
            info = ctx.index_info()
            if name == "dense":
                gpu = ctx.linearize(T0[:3, :3], T0[:3, 3], api.default_lin_params(0.5, 0), debug=True)
                ref = po.linearize(tree, src, T0[:3, :3], T0[:3, 3], po.default_lin_params(0.5, 0), debug=True)
                assert ref["n_eff"] > 3000
                assert_lin_equal(gpu, ref)
                assert_debug_equal(gpu, ref)
                ctx.set_source(src)
            res, logs = ctx.icp_run(T0, "Ours", cfg)
            runs[name] = (info.cell, int(info.n_cells), res, [(tuple(L.H_upper[:]), tuple(L.gradient[:]), L.effective_points, L.corr_pt_count) for L in logs])
            if name == "dense":
                ores, ologs = po.icp_run(tree, src, T0, "Ours", ocfg)
                assert_runs_equal(res, logs, ores, ologs)
                assert_cov_equal(res, ores)
                assert res.converged == 1 and logs[-1].trans_error_vs_gt < 0.01
        finally:
            ctx.close()
    assert runs["capped"][0] > 1.5 * runs["dense"][0] and runs["capped"][1] * 4 < runs["dense"][1]          # the cap did coarsen the grid ...
    assert runs["capped"][3] == runs["dense"][3]                                                           # ... and changed no bit of any sum


def test_the_montecarlo_job_of_one_rank_equals_the_python_driver():
    """dcreg_montecarlo_job (shard, run, gather over the ctx's communicator, statistics - one C-ABI call) with no communicator is a job of
    one rank: its records are bitwise those of dcreg_icp_run_montecarlo packed by dcreg_amd/montecarlo.py, its statistics
    icp_test_runner.cpp:604-664's; with a communicator of one rank (RCCL brought up) the same."""
    from dcreg_amd import montecarlo as mcm, pointshard
    pts = h.cylinder_cloud()
    cfg = api.default_config(search_radius=1.0, max_iterations=30, CONVERGENCE_THRESH_TRANS=1e-3, CONVERGENCE_THRESH_ROT=1e-5, KAPPA_TARGET=10.0,
                             STD_REG_GAMMA=100.0, use_weight_derivative=1, always_compute_schur=1)
    base = (0.2, 0.8, 0.5, h.deg2rad(0.1), h.deg2rad(0.1), h.deg2rad(2.0))
    ctx = api.Context(0)
    try:
        ctx.set_target(pts, 1.0); ctx.set_source(pts)
        n = 300
        want, wstats = mcm.run_montecarlo_native(ctx, "Ours", cfg, base, n, 2024, 0.5, np.deg2rad(2.0), slots=64)
        for with_comm in (False, True):
            if with_comm:
                pointshard.init_native_exchange(ctx)
            rec, st = ctx.montecarlo_job(base, 2024, n, 0.5, np.deg2rad(2.0), "Ours", cfg, slots=64)
            cols = [c for c in range(64) if c != mcm.R_TIME]                      # (wall time of a trial: not reproducible)
            assert np.array_equal(rec[:, cols], want[:, cols])
            assert st["total_runs"] == n == wstats["total_runs"] and st["converged_runs"] == wstats["converged_runs"] and st["ranks_seen"] == st["world"] == 1
            assert st["iterations_total"] == int(want[:, mcm.R_ITERS].sum()) and st["corr_num"] == wstats["corr_num"]
            for k in ("success_rate", "mean_trans_error", "std_trans_error", "min_trans_error", "max_trans_error", "mean_rot_error", "std_rot_error",
                      "mean_iterations", "mean_rmse", "mean_fitness"):
                assert np.isclose(st[k], wstats[k], rtol=1e-12, atol=1e-300), k
            got = ctx.comm_allgather(np.arange(5.0))
            assert got.shape == (1, 5) and np.array_equal(got[0], np.arange(5.0))
    finally:
        ctx.close()


# ---------------------------------------------------------------- k_lin in one-wave blocks (kernels.hpp k_lin<.., ONE>, k_sum_tiles)
def _same_sums(a, b):
    return (a["n_eff"] == b["n_eff"] and a["n_pt"] == b["n_pt"] and np.array_equal(a["H_upper"], b["H_upper"]) and np.array_equal(a["g"], b["g"])
            and a["sum_r2"] == b["sum_r2"] and a["sum_b2"] == b["sum_b2"])


@pytest.mark.parametrize("fast", [1, 0])
@pytest.mark.parametrize("scene", ["fixture", "lattice_dups", "planes_dense", "corridor_300k"])
def test_one_wave_blocks_are_invisible(scene, fast):
    """Walks that mix micrometre steps, centimetre steps and a jump (the scenes of round 5's pass tests: ties, duplicates, OUT points,
    dense cells; 30 to 1172 query blocks, i.e. ragged last chunks): with the linearisation kernel forced into one-wave blocks on every
    launch ("one_wave" = 2: a row per 64-point tile, k_sum_tiles behind the kernel), with and without certificates, and with the advance
    pass in front of it, the 31 sums are those of the four-wave launch bit for bit; the series reports which launches ran that way."""
    from test_gpu_round5 import _scene
    rng = np.random.default_rng(31)
    tgt, src, radius = _scene(scene, rng)
    prm = api.default_lin_params(radius, 1)
    ctxs = {}
    for name, opts in (("one", {"one_wave": 2, "advance": 0, "team_pass": 0}), ("one_all", {"one_wave": 2, "use_certificates": 0, "team_pass": 0}),
                       ("one_adv", {"one_wave": 2, "advance": 2, "team_pass": 0}),     # (launches behind the small-frame pass keep their four-wave blocks)
                       ("plain", {"one_wave": 0})):
        c = api.Context(0)
        c.set_option("fast_plane_fit", fast)
        for k, v in opts.items():
            c.set_option(k, v)
        c.set_option("record_launches", 1)
        c.set_target(tgt, radius); c.set_source(src)
        ctxs[name] = c
    T = np.eye(4)
    steps = [0.0, 1e-6, 1e-4, 3e-4, 1e-3, -1e-3, 2e-3, 1e-5, 4e-3, 6e-3, -6e-3, 1e-2, 1e-4, 3e-2, 0.2, 1e-3, 5e-4, 0.0]
    for k, sz in enumerate(steps):
        T = h.pose6d_matrix(sz * 0.6, -sz * 0.3, sz * 0.2, sz * 0.002, -sz * 0.001, sz * 0.004) @ T
        outs = {name: c.linearize(T[:3, :3], T[:3, 3], prm) for name, c in ctxs.items()}
        for name in ("one", "one_all", "one_adv"):
            assert _same_sums(outs[name], outs["plain"]), (scene, fast, k, name)
    ser = {name: c.launch_series(reset=True) for name, c in ctxs.items()}
    assert ser["one"]["one_wave"].all() and ser["one_all"]["one_wave"].all() and ser["one_adv"]["one_wave"].all() and not ser["plain"]["one_wave"].any()
    assert (ser["one_adv"]["advanced"][1:] == 1).all()
    for c in ctxs.values():
        c.close()


def test_one_wave_blocks_by_the_rule_in_whole_runs():
    """Engine level, 400 k corridor pair 0.35 m off (the pipelined engine: every launch but the first behind a gate; the run with the
    thresholds on calls its last queued launch off - k_lin and k_sum_tiles both return): the rule picks one-wave blocks for the first
    launches of a run (misalignment hint above 1.5 cells, most points searching) and four-wave blocks later; iteration by iteration H, g,
    counts and pose are bitwise those of runs with one-wave blocks everywhere and nowhere, and a second run from what the first left gives
    the same again."""
    tgt = h.scene_corridor(400_000, seed=9)
    src = (tgt + np.random.default_rng(10).normal(0, 0.01, tgt.shape)).astype(np.float32)
    T0 = h.pose6d_matrix(0.15, -0.2, 0.1, h.deg2rad(0.3), h.deg2rad(-0.2), h.deg2rad(0.6))
    logs, picked = {}, {}
    for name, ow in (("rule", 1), ("forced", 2), ("off", 0)):
        c = api.Context(0)
        c.set_option("one_wave", ow)
        c.set_option("record_launches", 1)
        c.set_target(tgt, 1.0); c.set_source(src)
        runs = []
        for rep, thr in enumerate((0.0, 0.0, None)):                  # two fixed-length runs, then one with the reference's thresholds on
            kw = {} if thr is None else {"CONVERGENCE_THRESH_ROT": 0.0, "CONVERGENCE_THRESH_TRANS": 0.0}
            cfg = api.default_config(search_radius=1.0, max_iterations=25, KAPPA_TARGET=10.0, STD_REG_GAMMA=100.0, use_weight_derivative=1,
                                     always_compute_schur=1, **kw)
            res, lg = c.icp_run(T0, "Ours", cfg)
            runs.append([(np.array(L.H_upper[:]), np.array(L.gradient[:]), L.effective_points, L.corr_pt_count, np.array(L.transform_matrix[:])) for L in lg[:res.iterations]])
        logs[name] = runs
        picked[name] = c.launch_series(reset=True)["one_wave"]
        c.close()
    for name in ("rule", "forced"):
        for rep in range(3):
            assert len(logs[name][rep]) == len(logs["off"][rep]) and len(logs["off"][rep]) >= 5
            for it, (x, y) in enumerate(zip(logs[name][rep], logs["off"][rep])):
                assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) and x[2] == y[2] and x[3] == y[3] and np.array_equal(x[4], y[4]), (name, rep, it)
    assert picked["off"].sum() == 0 and picked["forced"].all()
    n_rule = int(picked["rule"].sum())
    assert 3 <= n_rule <= 2 * len(picked["rule"]) // 3, picked["rule"]
    assert picked["rule"][0] == 1 and picked["rule"][24] == 0           # the first launch of the first run, its last one


def test_montecarlo_batches_in_one_wave_blocks_give_the_same_records():
    """The Monte-Carlo experiment (batched launches of 30-block poses, trials at every stage of their runs side by side) with the batches'
    linearisation kernel in one-wave blocks ("one_wave_batches" = 1, the default: a tile row per wave, k_sum_tiles with one block per pose
    behind it), in four-wave blocks (0), and with one-wave blocks forced for batches too small for the rule (12 slots, "one_wave" = 2):
    every trial's record - iterations, final pose, errors, H, mask - is bitwise the same."""
    from dcreg_amd import montecarlo as mcm
    pts = h.cylinder_cloud()
    cfg = api.default_config(search_radius=1.0, max_iterations=30, CONVERGENCE_THRESH_TRANS=1e-3, CONVERGENCE_THRESH_ROT=1e-5, KAPPA_TARGET=10.0,
                             STD_REG_GAMMA=100.0, use_weight_derivative=1, always_compute_schur=1)
    base = (0.2, 0.8, 0.5, h.deg2rad(0.1), h.deg2rad(0.1), h.deg2rad(2.0))
    cols = [c for c in range(64) if c != mcm.R_TIME]
    recs = {}
    for name, opts, slots in (("four", {"one_wave_batches": 0}, 128), ("one", {"one_wave_batches": 1}, 128), ("forced_small", {"one_wave": 2}, 12)):
        ctx = api.Context(0)
        try:
            for k, v in opts.items():
                ctx.set_option(k, v)
            ctx.set_target(pts, 1.0); ctx.set_source(pts)
            rec, st = ctx.montecarlo_job(base, 77, 400, 0.5, np.deg2rad(2.0), "Ours", cfg, slots=slots)
            recs[name] = rec[:, cols]
            assert st["total_runs"] == 400
        finally:
            ctx.close()
    assert np.array_equal(recs["one"], recs["four"]) and np.array_equal(recs["forced_small"], recs["four"])


def _window_pair(n_map=3_000_000, extent=90.0):
    tgt, src = h.scene_prior_map(n_map, extent=extent)
    return tgt, src, h.pose6d_matrix(**h.PK01_GT), h.pose6d_matrix(**h.PK01_INIT)


def _run_record(ctx, T0, cfg):
    res, logs = ctx.icp_run(T0, "Ours", cfg)
    return (tuple(res.R[:]), tuple(res.t[:]), res.iterations, res.converged, res.status,
            [(tuple(L.H_upper[:]), tuple(L.gradient[:]), L.effective_points, L.corr_pt_count, tuple(L.update_dx[:])) for L in logs])


@pytest.mark.timeout(900)
def test_the_window_index_of_a_large_map_is_invisible():
    """The window index (context.hpp: single-pose linearisations of a large map search an index over the map's points in a box around the
    transformed source): with it forced ("roi_index" 2) a walk of poses - steps inside the window, a jump that leaves it (a new window is
    built), the way back - gives bitwise the 31 sums of a context that searches the whole map; whole registrations (pipelined runs: gated
    launches, one of them with a margin of ZERO, so that nearly every pose leaves the window, the queued launch is called off and the window
    rebuilt) return bitwise the same poses, iteration counts and per-iteration sums; dcreg_knn and dcreg_p2p_error of the windowed context
    answer from the whole map."""
    tgt, src, gt, T0 = _window_pair()
    prm = api.default_lin_params(0.5, 0)
    cfg = api.default_config(search_radius=0.5, max_iterations=30, KAPPA_TARGET=10.0, STD_REG_GAMMA=100.0, CONVERGENCE_THRESH_ROT=1e-5,
                             CONVERGENCE_THRESH_TRANS=1e-3, use_weight_derivative=0, always_compute_schur=1, gt_matrix=gt.reshape(16))
    whole, win, win0 = api.Context(0), api.Context(0), api.Context(0)
    try:
        whole.set_option("roi_index", 0)
        win.set_option("roi_index", 2); win.set_option("roi_margin", 3.0)
        win0.set_option("roi_index", 2); win0.set_option("roi_margin", 0.0)
        for c in (whole, win, win0):
            c.set_target(tgt, 0.5); c.set_source(src)
        assert not win.roi_info()["active"]
        # a walk: inside the window, out of it, back
        steps = [(0, 0, 0, 0), (0.02, -0.01, 0.0, 0.001), (0.5, 0.4, 0.02, 0.01), (2.5, -2.0, 0.1, 0.02), (9.0, 7.0, 0.0, 0.05), (9.05, 7.02, 0.0, 0.051),
                 (0.02, -0.01, 0.0, 0.001), (-30.0, 25.0, 0.3, 0.2), (0, 0, 0, 0)]
        built = []
        for dx, dy, dz, yaw in steps:
            T = T0 @ h.pose6d_matrix(dx, dy, dz, 0.0, 0.0, yaw)
            a = whole.linearize(T[:3, :3], T[:3, 3], prm)
            b = win.linearize(T[:3, :3], T[:3, 3], prm)
            assert _same_sums(a, b), (dx, dy, dz, yaw, a["n_eff"], b["n_eff"])
            built.append(win.roi_info()["windows_built"])
        info = win.roi_info()
        assert info["active"] and 0 < info["points"] < len(tgt) and built[0] == 1 and built[2] == 1 and built[4] > built[2] and built[-1] > built[4]
        assert win.index_info().n_target == len(tgt) and whole.index_info().n_cells == win.index_info().n_cells       # (the info is the whole map's)
        # whole registrations (new frame each time, as the registration path runs: dcreg_set_source + run)
        for c in (whole, win, win0):
            c.set_source(src)
        r_whole, r_win, r_win0 = _run_record(whole, T0, cfg), _run_record(win, T0, cfg), _run_record(win0, T0, cfg)
        assert r_whole[4] == 0 and r_whole[2] > 5
        assert r_win == r_whole
        assert r_win0 == r_whole
        assert win0.roi_info()["windows_built"] > 3                     # margin 0: the window followed the pose through the run
        # a second run from the warm state of the first, and the other methods
        assert _run_record(win, T0, cfg) == _run_record(whole, T0, cfg)
        # everything else answers from the whole map
        q = (tgt[::40000] + np.float32(0.01)).astype(np.float32)
        ia, da = whole.knn(q, 5)
        ib, db = win.knn(q, 5)
        assert np.array_equal(ia, ib) and np.array_equal(da.view(np.uint32), db.view(np.uint32)) and not win.roi_info()["active"]
        assert win.p2p_error(T0, 1.0) == whole.p2p_error(T0, 1.0)
        b = win.linearize(T0[:3, :3], T0[:3, 3], prm)                   # ... and the next linearisation is back on the window
        assert win.roi_info()["active"] and _same_sums(whole.linearize(T0[:3, :3], T0[:3, 3], prm), b)
        # batches and dumps run on the whole map
        Rs = np.stack([T0[:3, :3], T0[:3, :3]]); ts = np.stack([T0[:3, 3], T0[:3, 3] + 0.01])
        ba, bb = whole.linearize_batch(Rs, ts, prm), win.linearize_batch(Rs, ts, prm)
        assert all(_same_sums(x, y) for x, y in zip(ba, bb)) and not win.roi_info()["active"]
    finally:
        for c in (whole, win, win0):
            c.close()


@pytest.mark.timeout(900)
def test_the_window_index_engages_by_itself_when_the_table_budget_binds():
    """Default rule ("roi_index" 1): a map whose dense cell table runs into "max_table_entries" - here a budget of 2^21 entries on a 3 M-point
    map, the situation of a 200 M-point map under the default 2^30 - registers frames on a window with the cells its density asks for; a map
    that fits its budget never builds one.  Same registration, bit for bit, either way."""
    tgt, src, gt, T0 = _window_pair()
    cfg = api.default_config(search_radius=0.5, max_iterations=30, KAPPA_TARGET=10.0, STD_REG_GAMMA=100.0, CONVERGENCE_THRESH_ROT=1e-5,
                             CONVERGENCE_THRESH_TRANS=1e-3, use_weight_derivative=0, always_compute_schur=1, gt_matrix=gt.reshape(16))
    fits, capped = api.Context(0), api.Context(0)
    try:
        capped.set_option("max_table_entries", 1 << 21)
        for c in (fits, capped):
            c.set_target(tgt, 0.5); c.set_source(src)
        ra, rb = _run_record(fits, T0, cfg), _run_record(capped, T0, cfg)
        assert ra == rb and ra[4] == 0 and ra[2] > 5
        ia, ib = fits.roi_info(), capped.roi_info()
        assert not ia["whole_map_capped"] and not ia["active"] and ia["windows_built"] == 0
        assert ib["whole_map_capped"] and ib["active"] and ib["windows_built"] == 1 and 0 < ib["points"] < len(tgt)
        assert ib["cell"] < 0.75 * capped.index_info().cell          # the window's cells are the density's, the whole map's the budget's
        # the next frames of the drive reuse the window
        for k in range(3):
            T = T0 @ h.pose6d_matrix(0.3 * (k + 1), 0.1 * k, 0.0, 0.0, 0.0, 0.002 * k)
            for c in (fits, capped):
                c.set_source(src)
            assert _run_record(fits, T, cfg) == _run_record(capped, T, cfg)
        assert capped.roi_info()["windows_built"] == 1
    finally:
        fits.close(); capped.close()
