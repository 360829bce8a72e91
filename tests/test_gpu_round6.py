"""GPU tests added in round 6 (run with -m gpu on an MI355X; everything through the C-ABI):
  * clouds with non-finite coordinates are refused, source and target alike (NaN as well as infinity);
  * dcreg_reset_warm_state(ctx, -1) drops the context's own neighbour state: the next run is bitwise the run of a fresh context (the
    reference's fresh ICPContext per run, icp_test_runner.cpp:408-409 - what bench.py's `cold_run` times);
  * small frames from host buffers go through the context's pinned block: the caller's buffer may be overwritten as soon as
    dcreg_set_source returns."""
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
