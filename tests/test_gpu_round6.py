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


# ---------------------------------------------------------------- the wave-cooperative tile search (search.hpp tile_search6)
from oracle import pyoracle as po          # noqa: E402  (the checker)
from test_gpu_parity import assert_lin_equal, assert_debug_equal          # noqa: E402

TILE_SCENES = {
    # name: target, radius, offsets (metres) of the source from the target - from aligned to a cell-dozen away
    "corridor_300k": (lambda: h.scene_corridor(300_000, seed=5), 1.0, (0.0, 0.05, 0.3, 0.8)),
    "cylinder_60k": (lambda: h.scene_cylinder(60_000, seed=8, noise=0.01), 1.0, (0.02, 0.4)),
    "planes_r04": (lambda: h.scene_planes(80_000, seed=4), 0.4, (0.0, 0.15, 0.35)),
    "fixture": (lambda: h.cylinder_cloud(), 1.0, (0.1, 0.6)),
}


@pytest.mark.parametrize("fast", [1, 0])
@pytest.mark.parametrize("scene", list(TILE_SCENES))
def test_tile_search_matches_the_oracle_point_by_point(scene, fast):
    """Every dense wave searches over a shared candidate tile ("tile_search" = 2: whenever the tile can be held): neighbour indices, float
    distances as bit patterns and gate flags against the oracle's kd-tree, at offsets from aligned to most of the search radius."""
    gen, radius, offsets = TILE_SCENES[scene]
    tgt = gen()
    rng = np.random.default_rng(3)
    src = (tgt[:: 2] + rng.normal(0, 0.005, tgt[::2].shape)).astype(np.float32)
    tree = po.KdTree(tgt)
    ctx = api.Context(0)
    try:
        ctx.set_option("fast_plane_fit", fast)
        ctx.set_option("tile_search", 2); ctx.set_option("tile_max_pts", 1 << 20); ctx.set_option("count_searches", 1)
        ctx.set_target(tgt, radius); ctx.set_source(src)
        used = 0
        for off in offsets:
            T = h.pose6d_matrix(off * 0.7, -off * 0.5, off * 0.5, 0.002, -0.001, 0.004 * (1 + off))
            ctx.launch_stats(reset=True)
            gpu = ctx.linearize(T[:3, :3], T[:3, 3], api.default_lin_params(radius, 1), debug=True)
            used += ctx.launch_stats(reset=True)["points_tile"]
            ref = po.linearize(tree, src, T[:3, :3], T[:3, 3], po.default_lin_params(radius, 1), debug=True)
            assert_lin_equal(gpu, ref)
            assert_debug_equal(gpu, ref)
        assert used > len(src)            # the tile search did carry (most of) these launches
    finally:
        ctx.close()


@pytest.mark.parametrize("scene", ["corridor_300k", "cylinder_60k", "lattice_dups", "planes_r04"])
def test_tile_search_is_invisible(scene):
    """Walks that mix micrometre steps, decimetre jumps and a metre jump: tile search forced, by the rule, and off give bitwise the same 31
    sums at every step - whatever the states hold and whoever searched (history independence); ties (the lattice) fall back to the exact
    keys of the lock-step search."""
    if scene == "lattice_dups":
        g = np.arange(0, 14, dtype=np.float32) * 0.3
        tgt = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
        tgt, radius = np.concatenate([tgt, tgt[::7]]), 0.7
        src = (tgt[::3] + np.float32(0.11)).astype(np.float32)
    else:
        gen, radius, _ = TILE_SCENES[scene]
        tgt = gen()
        src = (tgt[::2] + np.random.default_rng(31).normal(0, 0.004, tgt[::2].shape)).astype(np.float32)
    prm = api.default_lin_params(radius, 1)
    ctxs = {}
    for name, opts in (("forced", {"tile_search": 2, "tile_max_pts": 1 << 20}), ("rule", {"tile_search": 1}), ("off", {"tile_search": 0}),
                       ("forced_nocert", {"tile_search": 2, "use_certificates": 0})):
        c = api.Context(0)
        for k, v in opts.items():
            c.set_option(k, v)
        c.set_option("count_searches", 1)
        c.set_target(tgt, radius); c.set_source(src)
        ctxs[name] = c
    T = np.eye(4)
    steps = [0.0, 1e-6, 1e-3, 0.3, -0.25, 0.02, 0.6, -0.6, 1e-4, 0.1, 1.0, -1.1, 5e-3, 0.0]
    for k, sz in enumerate(steps):
        T = h.pose6d_matrix(sz * 0.6, -sz * 0.5, sz * 0.4, sz * 0.002, -sz * 0.001, sz * 0.004) @ T
        outs = {name: c.linearize(T[:3, :3], T[:3, 3], prm) for name, c in ctxs.items()}
        for name in ("forced", "rule", "forced_nocert"):
            assert _same_sums(outs[name], outs["off"]), (scene, name, k)
    st = {name: c.launch_stats(reset=True) for name, c in ctxs.items()}
    assert st["off"]["points_tile"] == 0
    if scene != "lattice_dups":
        assert st["forced"]["points_tile"] > 0
    if scene == "corridor_300k":
        assert st["rule"]["points_tile"] > 0            # (the decimetre and metre jumps: dense waves far from their surface)
    for c in ctxs.values():
        c.close()


def _same_sums(a, b):
    return (a["n_eff"] == b["n_eff"] and a["n_pt"] == b["n_pt"] and np.array_equal(a["H_upper"], b["H_upper"]) and np.array_equal(a["g"], b["g"])
            and a["sum_r2"] == b["sum_r2"] and a["sum_b2"] == b["sum_b2"])


def test_tile_search_in_whole_pipelined_runs():
    """Engine level: 30-iteration runs of a 400 k corridor pair from the bench's initial pose (gated, pipelined launches), tile search by the
    rule / forced / off: every iteration's H, g, counts and pose bitwise the same; the rule used it in the first launches only."""
    tgt = h.scene_corridor(400_000, seed=9)
    src = (tgt + np.random.default_rng(10).normal(0, 0.01, tgt.shape)).astype(np.float32)
    T0 = h.pose6d_matrix(0.05, -0.08, 0.03, h.deg2rad(0.2), h.deg2rad(-0.1), h.deg2rad(0.5))
    cfg = api.default_config(search_radius=1.0, max_iterations=30, KAPPA_TARGET=10.0, STD_REG_GAMMA=100.0, CONVERGENCE_THRESH_ROT=0.0,
                             CONVERGENCE_THRESH_TRANS=0.0, use_weight_derivative=1, always_compute_schur=1)
    logs, tiles = {}, {}
    for name, opts in (("rule", {"tile_search": 1}), ("forced", {"tile_search": 2, "tile_max_pts": 1 << 20}), ("off", {"tile_search": 0})):
        c = api.Context(0)
        for k, v in opts.items():
            c.set_option(k, v)
        c.set_option("count_searches", 1)
        c.set_target(tgt, 1.0); c.set_source(src)
        runs = []
        for rep in range(2):                       # the second run starts from the first one's converged state
            res, lg = c.icp_run(T0, "Ours", cfg)
            runs.append([(np.array(L.H_upper[:]), np.array(L.gradient[:]), L.effective_points, L.corr_pt_count, np.array(L.transform_matrix[:])) for L in lg[:res.iterations]])
        logs[name] = runs
        tiles[name] = c.launch_stats(reset=True)
        c.close()
    for name in ("rule", "forced"):
        for rep in range(2):
            assert len(logs[name][rep]) == len(logs["off"][rep]) == 30
            for it, (x, y) in enumerate(zip(logs[name][rep], logs["off"][rep])):
                assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) and x[2] == y[2] and x[3] == y[3] and np.array_equal(x[4], y[4]), (name, rep, it)
    assert tiles["off"]["points_tile"] == 0 and tiles["forced"]["points_tile"] >= tiles["rule"]["points_tile"] > 0
    assert tiles["rule"]["points_tile"] < tiles["rule"]["points_searched"]
