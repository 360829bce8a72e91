"""GPU tests added in round 5 (run with -m gpu on an MI355X; everything through the C-ABI):
  * the advance pass (kernels.hpp k_advance: the searches and refits of a launch in dense waves, in front of the linearisation
    kernel) changes no sum - walks on scenes with ties, duplicates, OUT points and dense cells, whole engine runs, gated launches;
  * the plane fit of the parity mode (fast_plane_fit = 0) takes its rows in distance order, like the reference."""
import numpy as np
import pytest

import helpers as h
from dcreg_amd import api

pytestmark = pytest.mark.gpu


def _same_sums(a, b):
    return (a["n_eff"] == b["n_eff"] and a["n_pt"] == b["n_pt"] and np.array_equal(a["H_upper"], b["H_upper"]) and np.array_equal(a["g"], b["g"])
            and a["sum_r2"] == b["sum_r2"] and a["sum_b2"] == b["sum_b2"])


def _scene(scene, rng):
    if scene == "cylinder_60k":
        tgt, radius = h.scene_cylinder(60_000, seed=8, noise=0.01), 1.0
    elif scene == "fixture":
        tgt, radius = h.cylinder_cloud(), 1.0
    elif scene == "planes_dense":
        tgt, radius = h.scene_planes(80_000, seed=4), 0.4
    elif scene == "corridor_300k":
        tgt, radius = h.scene_corridor(300_000, seed=5), 1.0
    else:
        g = np.arange(0, 14, dtype=np.float32) * 0.3
        tgt = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
        tgt, radius = np.concatenate([tgt, tgt[::7]]), 0.7
    src = (tgt[::2] + rng.normal(0, 0.004, tgt[::2].shape)).astype(np.float32)
    if scene == "lattice_dups":
        src = (tgt[::3] + np.float32(0.11)).astype(np.float32)
    if scene == "corridor_300k":
        src = (tgt + rng.normal(0, 0.01, tgt.shape)).astype(np.float32)
    return tgt, src, radius


@pytest.mark.parametrize("fast", [1, 0])
@pytest.mark.parametrize("scene", ["cylinder_60k", "fixture", "lattice_dups", "planes_dense", "corridor_300k"])
def test_advance_pass_is_invisible(scene, fast):
    """Walks that mix micrometre steps, centimetre steps and a jump: with the advance pass forced on every launch that can take it
    ("advance" = 2), never (0), and with certificates off (every point searched inside the linearisation kernel), the 31 sums agree
    bit for bit at every step; the pass did run, and it reports its searches through the launch's own count slots."""
    rng = np.random.default_rng(31)
    tgt, src, radius = _scene(scene, rng)
    prm = api.default_lin_params(radius, 1)
    ctxs = {}
    for name, opts in (("adv", {"advance": 2, "team_pass": 0}), ("team", {"advance": 0, "team_pass": 2}), ("plain", {"advance": 0, "team_pass": 0}),
                       ("all", {"use_certificates": 0, "advance": 0, "team_pass": 0})):
        c = api.Context(0)
        c.set_option("fast_plane_fit", fast)
        for k, v in opts.items():
            c.set_option(k, v)
        c.set_option("record_launches", 1)
        c.set_target(tgt, radius); c.set_source(src)
        ctxs[name] = c
    T = np.eye(4)
    steps = [0.0, 1e-6, 1e-4, 3e-4, 1e-3, -1e-3, 2e-3, 1e-5, 4e-3, 6e-3, -6e-3, 1e-2, 1e-4, 3e-2, 0.2, 1e-3, 5e-4, 0.0]
    ran, searched_adv, searched_team, searched_plain = 0, 0, 0, 0
    for k, sz in enumerate(steps):
        T = h.pose6d_matrix(sz * 0.6, -sz * 0.3, sz * 0.2, sz * 0.002, -sz * 0.001, sz * 0.004) @ T
        outs = {name: c.linearize(T[:3, :3], T[:3, 3], prm) for name, c in ctxs.items()}
        assert _same_sums(outs["adv"], outs["plain"]) and _same_sums(outs["adv"], outs["all"]), (scene, fast, k)
        assert _same_sums(outs["team"], outs["plain"]), (scene, fast, k)
        sa, st, sp = ctxs["adv"].launch_series(reset=True), ctxs["team"].launch_series(reset=True), ctxs["plain"].launch_series(reset=True)
        ctxs["all"].launch_series(reset=True)
        assert len(sa["ms"]) == 1 and sa["advanced"][0] == (1 if k > 0 else 0) and sp["advanced"][0] == 0 and st["advanced"][0] == (2 if k > 0 else 0)
        ran += int(sa["advanced"][0])
        if k > 0:
            # (the contexts need not search the same points: the certificates a search leaves depend on HOW it was carried out - the
            #  team searches know the seventh distance exactly, the lock-step search a lower bound of it)
            searched_adv += int(sa["searched"][0]); searched_plain += int(sp["searched"][0]); searched_team += int(st["searched"][0])
    assert ran == len(steps) - 1 and searched_adv > 0 and searched_plain > 0 and searched_team > 0
    for c in ctxs.values():
        c.close()


def test_advance_pass_in_whole_runs_and_behind_the_gate():
    """Engine level: 30-iteration runs of a 400 k corridor pair (the pipelined engine queues every launch behind a gate) with the pass
    chosen by the host's rule made to fire ("advance_min_blocks" = 1), forced, and off: every iteration's H, g, counts and pose are
    bitwise the same; the rule did pick the pass for some launches and not for others."""
    tgt = h.scene_corridor(400_000, seed=9)
    src = (tgt + np.random.default_rng(10).normal(0, 0.01, tgt.shape)).astype(np.float32)
    T0 = h.pose6d_matrix(0.05, -0.08, 0.03, h.deg2rad(0.2), h.deg2rad(-0.1), h.deg2rad(0.5))
    cfg = api.default_config(search_radius=1.0, max_iterations=30, KAPPA_TARGET=10.0, STD_REG_GAMMA=100.0, CONVERGENCE_THRESH_ROT=0.0,
                             CONVERGENCE_THRESH_TRANS=0.0, use_weight_derivative=1, always_compute_schur=1)
    logs, picked = {}, {}
    for name, opts in (("rule", {"advance": 1, "advance_min_blocks": 1, "team_pass": 0}), ("forced", {"advance": 2, "team_pass": 0}), ("off", {"advance": 0, "team_pass": 0})):
        c = api.Context(0)
        for k, v in opts.items():
            c.set_option(k, v)
        c.set_option("record_launches", 1)
        c.set_target(tgt, 1.0); c.set_source(src)
        runs = []
        for rep in range(2):                       # the second run starts from the first one's converged state
            res, lg = c.icp_run(T0, "Ours", cfg)
            runs.append([(np.array(L.H_upper[:]), np.array(L.gradient[:]), L.effective_points, L.corr_pt_count, np.array(L.transform_matrix[:])) for L in lg[:res.iterations]])
        logs[name] = runs
        picked[name] = c.launch_series(reset=True)["advanced"]
        c.close()
    for name in ("rule", "forced"):
        for rep in range(2):
            assert len(logs[name][rep]) == len(logs["off"][rep]) == 30
            for it, (x, y) in enumerate(zip(logs[name][rep], logs["off"][rep])):
                assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) and x[2] == y[2] and x[3] == y[3] and np.array_equal(x[4], y[4]), (name, rep, it)
    assert picked["off"].sum() == 0 and picked["forced"].sum() >= 58
    assert 0 < picked["rule"].sum() < 50, picked["rule"]


def test_small_frame_registration_with_the_team_pass():
    """The reference's own workload (icp_test_runner.cpp:442-461): an 8 k-point frame registered against a 200 k-point map, from host
    buffers, to convergence.  With the small-frame pass by the host's rule (the default), forced and off: iteration for
    iteration the same H, g, counts and pose, bit for bit."""
    tgt, src = h.scene_parkinglot()
    gt, T0 = h.pose6d_matrix(**h.PK01_GT), h.pose6d_matrix(**h.PK01_INIT)
    cfg = api.default_config(search_radius=0.5, max_iterations=30, KAPPA_TARGET=10.0, STD_REG_GAMMA=100.0, CONVERGENCE_THRESH_ROT=1e-5,
                             CONVERGENCE_THRESH_TRANS=1e-3, use_weight_derivative=0, always_compute_schur=1, gt_matrix=gt.reshape(16))
    logs, picked = {}, {}
    for name, opts in (("rule", {}), ("forced", {"team_pass": 2}), ("off", {"team_pass": 0})):
        c = api.Context(0)
        for k, v in opts.items():
            c.set_option(k, v)
        c.set_option("record_launches", 1)
        c.set_target(tgt, 0.5)
        runs = []
        for rep in range(2):
            c.set_source(src)
            res, lg = c.icp_run(T0, "Ours", cfg)
            assert res.converged == 1
            runs.append([(np.array(L.H_upper[:]), np.array(L.gradient[:]), L.effective_points, L.corr_pt_count, np.array(L.transform_matrix[:])) for L in lg[:res.iterations]])
        logs[name] = runs
        picked[name] = c.launch_series(reset=True)["advanced"]
        c.close()
    for name in ("rule", "forced"):
        for rep in range(2):
            assert len(logs[name][rep]) == len(logs["off"][rep])
            for it, (x, y) in enumerate(zip(logs[name][rep], logs["off"][rep])):
                assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) and x[2] == y[2] and x[3] == y[3] and np.array_equal(x[4], y[4]), (name, rep, it)
    assert picked["off"].sum() == 0
    n_it = len(logs["off"][0])
    assert (picked["forced"] == 2).sum() >= 2 * (n_it - 1) - 1, picked["forced"]
    assert (picked["rule"] == 2).sum() >= 2 * (n_it - 8), picked["rule"]      # (every launch expected to search most of the frame)


def test_parity_fit_takes_its_rows_in_distance_order():
    """fast_plane_fit = 0 is the reference's factorisation step for step, rows in DISTANCE order (icp_test_runner.cpp:1733-1747): on a
    scene of nearly collinear / rank-deficient neighbourhoods the normals agree with the oracle's (which fits in distance order) far
    closer than a row permutation would leave them, and walking keeps the sums those of a context that searches everything."""
    from oracle import pyoracle as po
    rng = np.random.default_rng(77)
    # points on a few lines and planes: many neighbourhoods are rank 2 (collinear in projection)
    t = rng.uniform(0, 10, 6000)
    lines = np.stack([t, 0.02 * rng.normal(size=t.shape), np.round(rng.uniform(0, 3, t.shape))], 1)
    plane = np.stack([rng.uniform(0, 10, 6000), rng.uniform(0, 10, 6000), 5.0 + 1e-4 * rng.normal(size=6000)], 1)
    tgt = np.concatenate([lines, plane]).astype(np.float32)
    src = (tgt[::3] + rng.normal(0, 0.01, tgt[::3].shape)).astype(np.float32)
    T = h.pose6d_matrix(0.01, -0.02, 0.015, h.deg2rad(0.1), 0.0, h.deg2rad(-0.1))
    prm = api.default_lin_params(1.0, 1)
    c = api.Context(0)
    c.set_option("fast_plane_fit", 0)
    c.set_target(tgt, 1.0); c.set_source(src)
    d = c.linearize(T[:3, :3], T[:3, 3], prm, debug=True)
    tree = po.KdTree(tgt)
    o = po.linearize(tree, src, T[:3, :3], T[:3, 3], po.default_lin_params(1.0, 1), debug=True)
    assert np.array_equal(d["flag"], o["flag"])
    ok = (d["flag"] == 1) | (d["flag"] == 4)
    assert ok.sum() > 500
    assert np.max(np.abs(d["normal"][ok] - o["normal"][ok])) < 1e-9
    # a walk with the stored planes in use equals a context that refits everything
    f = api.Context(0)
    f.set_option("fast_plane_fit", 0); f.set_option("use_certificates", 0)
    f.set_target(tgt, 1.0); f.set_source(src)
    for sz in (0.0, 1e-5, 1e-3, -1e-3, 5e-3, 1e-4):
        T = h.pose6d_matrix(sz, -sz * 0.5, sz * 0.2, sz * 0.01, 0.0, -sz * 0.01) @ T
        assert _same_sums(c.linearize(T[:3, :3], T[:3, 3], prm), f.linearize(T[:3, :3], T[:3, 3], prm)), sz
    c.close(); f.close()


def test_two_rccl_ranks_on_one_device_are_refused_by_rccl_itself():
    """The one-GPU box cannot exercise RCCL with more than one rank: RCCL (like NCCL) refuses a communicator whose ranks share a device
    ("invalid usage": duplicate GPU).  This test pins that reason down - two processes, backend nccl, both on cuda:0, 127.0.0.1
    rendezvous (scripts/rccl_two_ranks_probe.py) - so that DESIGN.md's "RCCL with N > 1 ranks has never run" stays a statement about
    the hardware available, not about the code: the two-rank paths of the library run over gloo on this box (test_bench_launcher,
    test_two_ranks_...), the RCCL path with a communicator of one rank (test_native_rccl_exchange_world_of_one).  Should RCCL ever
    accept the shared device, both ranks must then complete their all_gather."""
    import os, subprocess, sys
    out = subprocess.run([sys.executable, os.path.join(h.REPO, "scripts", "rccl_two_ranks_probe.py")], capture_output=True, text=True, timeout=300).stdout
    ok = out.count("torch nccl all_gather ok")
    refused = out.count("invalid usage") + out.count("Duplicate GPU")
    assert ok == 2 or (ok == 0 and refused >= 2), out[-2000:]


def test_bounded_hunt_for_the_advance_passes():
    """scripts/fuzz_passes.py with a fixed seed inside the suite: 40 random scenes (five kinds incl. a lattice with duplicates; radii 0.3-2 m
    against cells of half to three neighbour spacings: balls of one to a dozen cells; either plane fit; with / without the empty-space
    field) x 6 steps from 10 um to decimetres: dense pass forced == team pass forced == no pass, bit for bit, and == oracle."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("fuzz_passes", os.path.join(h.REPO, "scripts", "fuzz_passes.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    assert mod.run(40, 20260928, verbose=False) == 0


@pytest.mark.parametrize("team", [0, 2])
def test_small_launches_gated_in_their_first_kernel(team):
    """A pipelined launch of at most 64 query blocks waits for its pose inside its first kernel (k_lin, or k_advance_team when the teams
    run) instead of behind the gate kernel.  The protocol of test_gated_launches_equal_blocking_ones on the fixture, with the gate inside
    and - option "gate_in_kernel" 0 - behind k_gate: bitwise the blocking results; a launch that is called off leaves results and state
    untouched; called off and the next one published at once (the polling wave may only start when the record already carries the later
    number); a context destroyed with a gate still waiting."""
    tgt = h.cylinder_cloud()
    src = tgt[::3]
    prm = api.default_lin_params(1.0, 1)
    poses = [h.pose6d_matrix(0.05 * k, -0.03 * k, 0.02, h.deg2rad(0.3 * k), 0.0, h.deg2rad(-0.2 * k)) for k in range(6)]
    ref = api.Context(0)
    ref.set_option("team_pass", 0)
    ref.set_target(tgt, 1.0); ref.set_source(src)
    want = [ref.linearize(T[:3, :3], T[:3, 3], prm) for T in poses]
    ref.close()
    for inside in (1, 0):
        c = api.Context(0)
        c.set_option("team_pass", team); c.set_option("gate_in_kernel", inside); c.set_option("record_launches", 1)
        c.set_target(tgt, 1.0); c.set_source(src)
        slot = 0
        c.linearize_begin(poses[0][:3, :3], poses[0][:3, 3], prm, slot=slot)
        got = []
        for k in range(len(poses)):
            last = k + 1 == len(poses)
            if not last:
                c.linearize_gated_begin(prm, slot=slot ^ 1)
            got.append(c.linearize_end(slot=slot))
            if not last:
                c.gate_open(poses[k + 1][:3, :3], poses[k + 1][:3, 3])
                slot ^= 1
        for a, b in zip(got, want):
            assert _same_sums(a, b), (team, inside)
        assert (c.launch_series(reset=True)["advanced"][1:] == team).all()
        c.linearize_gated_begin(prm, slot=1)
        c.gate_abort()
        again = c.linearize(poses[-1][:3, :3], poses[-1][:3, 3], prm)
        assert _same_sums(again, want[-1])
        for _ in range(20):
            c.linearize_gated_begin(prm, slot=1)
            c.gate_abort()
            c.linearize_gated_begin(prm, slot=1)
            c.gate_open(poses[-2][:3, :3], poses[-2][:3, 3])
            out = c.linearize_end(slot=1)
            assert _same_sums(out, want[-2])
            c.linearize_gated_begin(prm, slot=0)
            c.gate_open(poses[-1][:3, :3], poses[-1][:3, 3])
            assert _same_sums(c.linearize_end(slot=0), want[-1])
        c.linearize_gated_begin(prm, slot=1)
        c.close()                                                   # destroys the context with the gate still waiting
