"""The device ALGORITHM on the CPU: dcreg_amd/csrc/device/search.hpp (the per-thread functions of the HIP kernels: warm
bound, grid ring walk with the empty-space field, exact tie handling, plane fit, gates, row) compiled for the host by
tests/host_emul/ and compared with the oracle -- bit-exact neighbour indices / float distances / gate flags, sums to
rounding.  This is test infrastructure (like oracle/): the product has no host path; the same comparisons run on the GPU in
tests/test_gpu_parity.py.  What it buys: the exactness of every search change is checked here, without a GPU, on the cases
that break grid searches -- distance ties, duplicates, queries outside the grid, empty space between clusters, many-cell
misalignment, cloud borders."""
import numpy as np
import pytest

import emul
import helpers as h
from oracle import pyoracle as po


def assert_same(e, ref, normals_atol=1e-8, h_rtol=1e-11):
    assert e["n_eff"] == ref["n_eff"] and e["n_pt"] == ref["n_pt"]
    assert np.array_equal(e["flag"], ref["flag"])
    ok = (ref["flag"] != 0) & (e["nn_idx"][:, 0] != -2)        # (-2: the point was linearised on its stored plane, no neighbour list rebuilt)
    assert np.array_equal(e["nn_idx"][ok], ref["nn_idx"][ok])
    assert np.array_equal(e["nn_d2"][ok].view(np.uint32), ref["nn_d2"][ok].view(np.uint32))
    passed = (ref["flag"] == 1) | (ref["flag"] == 4)
    assert np.allclose(e["normal"][passed], ref["normal"][passed], rtol=0, atol=normals_atol)
    assert np.allclose(e["r"][passed], ref["r"][passed], rtol=0, atol=normals_atol)
    assert np.allclose(e["s"][passed], ref["s"][passed], rtol=0, atol=normals_atol)
    if ref["n_eff"] > 0:
        assert h.rel_err(e["H_upper"], ref["H_upper"]) < h_rtol and h.rel_err(e["g"], ref["g"]) < max(1e-9, h_rtol)


@pytest.mark.parametrize("init,wd", [(h.RELEASE_INIT, 0), (h.PAPER_INIT, 1)])
@pytest.mark.parametrize("fast", [False, True])
def test_fixture_matches_oracle_and_golden(init, wd, fast):
    pts = h.cylinder_cloud()
    idx, src, tree = emul.Index(pts, 1.0), emul.Source(pts), po.KdTree(pts)
    T0 = h.pose6d_matrix(**init)
    ref = po.linearize(tree, pts, T0[:3, :3], T0[:3, 3], po.default_lin_params(1.0, wd), debug=True)
    for _ in range(2):                                   # cold, then bounded by its own previous result
        e = emul.linearize(idx, src, T0[:3, :3], T0[:3, 3], wd=wd, fast=fast, debug=True)
        assert_same(e, ref, normals_atol=1e-10)
    assert e["n_eff"] == (871 if wd == 0 else 197)       # committed traces, iteration 0 (incl. the 751 rank-2 neighbourhoods)


SCENES = {
    "cylinder_20k": lambda: (h.scene_cylinder(20_000, seed=1, noise=0.01), 1.0),
    "planes_20k_r05": lambda: (h.scene_planes(20_000, seed=2), 0.5),
    "corridor_60k": lambda: (h.scene_corridor(60_000, seed=3, length=40.0), 1.0),
    "sparse_3k": lambda: (h.scene_cylinder(3_000, seed=4), 1.0),
}


@pytest.mark.parametrize("name", list(SCENES))
def test_synthetic_scenes_and_pose_sequences(name):
    """Small, large (many cells: ring walk, face culling) and absurd misalignments in sequence, warm state carried along."""
    tgt, radius = SCENES[name]()
    rng = np.random.default_rng(11)
    src = tgt[rng.permutation(len(tgt))[: len(tgt) // 2]].copy()
    src += rng.normal(0, 0.01, src.shape).astype(np.float32)
    idx, S, tree = emul.Index(tgt, radius), emul.Source(src), po.KdTree(tgt)
    poses = [h.pose6d_matrix(0.05, -0.08, 0.03, h.deg2rad(0.2), h.deg2rad(-0.1), h.deg2rad(0.5)),
             h.pose6d_matrix(0.4, -0.5, 0.3, h.deg2rad(1.0), h.deg2rad(-2.0), h.deg2rad(3.0)),       # several cells off
             h.pose6d_matrix(0.0, 0.0, 0.0, 0.0, 0.0, 0.0),
             h.pose6d_matrix(300.0, 0.0, 0.0, 0.0, 0.0, 0.0),                                        # far outside the grid
             h.pose6d_matrix(-0.3, 0.2, 0.6, h.deg2rad(-1.0), h.deg2rad(0.5), h.deg2rad(-2.0))]
    for k, T in enumerate(poses):
        ref = po.linearize(tree, src, T[:3, :3], T[:3, 3], po.default_lin_params(radius, k % 2), debug=True)
        e = emul.linearize(idx, S, T[:3, :3], T[:3, 3], radius=radius, wd=k % 2, fast=True, debug=True)
        assert_same(e, ref)
    # certificates only spare searches: along a walk that mixes tiny steps, jumps and a trip far outside the grid, charging them at
    # every launch ("cert"), never ("full") and by the product's rule (None) give the same per-point results and bitwise the same sums
    walk = [poses[1], poses[1] @ h.pose6d_matrix(1e-4, -2e-4, 1e-4, 1e-6, -2e-6, 1e-6), poses[1] @ h.pose6d_matrix(2e-3, 1e-3, -1e-3, 1e-5, 2e-5, -1e-5),
            poses[2], poses[2] @ h.pose6d_matrix(0.02, 0.0, 0.01, 0.0, 1e-4, 0.0), poses[4], poses[3], poses[0], poses[0]]
    runs = {}
    for plan in ("full", None, "cert"):
        Sw = emul.Source(src)
        runs[plan] = [emul.linearize(idx, Sw, T[:3, :3], T[:3, 3], radius=radius, wd=1, plan=plan, debug=(plan == "cert")) for T in walk]
    assert sum(r["plan"] == "cert" for r in runs[None]) >= 3 and sum(r["searched"] for r in runs["cert"]) < sum(r["searched"] for r in runs["full"])
    for plan in (None, "cert"):
        for x, y in zip(runs[plan], runs["full"]):
            assert x["n_eff"] == y["n_eff"] and np.array_equal(x["H_upper"], y["H_upper"]) and np.array_equal(x["g"], y["g"])
    for T, x in zip(walk, runs["cert"]):
        assert_same(x, po.linearize(tree, src, T[:3, :3], T[:3, 3], po.default_lin_params(radius, 1), debug=True))
    # the warm bound and the empty-space field only prune: cold / no-field searches give bitwise the same sums
    T = poses[1]
    a = emul.linearize(idx, S, T[:3, :3], T[:3, 3], radius=radius, wd=1)
    S2 = emul.Source(src)
    b = emul.linearize(emul.Index(tgt, radius, gap_field=False), S2, T[:3, :3], T[:3, 3], radius=radius, wd=1, warm=False)
    assert a["n_eff"] == b["n_eff"] and np.array_equal(a["H_upper"], b["H_upper"]) and np.array_equal(a["g"], b["g"])


def test_row_sweep_and_ring_walk_agree_bit_for_bit():
    """k_lin's searches sweep the occupied rows of their ball (search.hpp knn_shells<.., true>); dcreg_knn and -DDCREG_RING_WALK builds
    walk rings.  Same neighbours, same certificates (the whole state, bit for bit), same sums - aligned, many cells off, far outside the
    grid, with a warm bound and without, on a lattice with duplicated points (exact ties: the 64-bit-key search) and far from the origin."""
    rng = np.random.default_rng(5)
    g = np.arange(0, 10, dtype=np.float32) * 0.3
    lattice = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    cases = {name: make() for name, make in SCENES.items()}
    cases["lattice_dups"] = (np.concatenate([lattice, lattice[::7]]), 0.7)
    cases["far_corridor"] = ((h.scene_corridor(8000, seed=6, length=20.0).astype(np.float64) + np.array([3.0e4, -2.0e4, 500.0])).astype(np.float32), 0.8)
    poses = [h.pose6d_matrix(0.4, -0.5, 0.3, h.deg2rad(1.0), h.deg2rad(-2.0), h.deg2rad(3.0)),
             h.pose6d_matrix(0.35, -0.45, 0.3, h.deg2rad(1.0), h.deg2rad(-1.8), h.deg2rad(2.5)),
             h.pose6d_matrix(0.0, 0.0, 0.0, 0.0, 0.0, 0.0),
             h.pose6d_matrix(0.0, 0.0, 1e-4, 0.0, 0.0, 0.0),
             h.pose6d_matrix(300.0, 0.0, 0.0, 0.0, 0.0, 0.0),
             h.pose6d_matrix(-0.3, 0.2, 0.6, h.deg2rad(-1.0), h.deg2rad(0.5), h.deg2rad(-2.0))]
    swept = 0
    for name, (tgt, radius) in cases.items():
        src = (tgt[::2] + rng.normal(0, 0.01, tgt[::2].shape)).astype(np.float32)
        c = src.astype(np.float64).mean(axis=0)
        shift = np.eye(4); shift[:3, 3] = c
        for cell in (0.0, radius / 7.3):                            # the density-adapted cell, and one that makes every ball span many cells
            idx = emul.Index(tgt, radius, cell=cell)
            out = {}
            for sweep in (True, False):
                idx.set_sweep(sweep)
                S = emul.Source(src)
                out[sweep] = []
                for T in poses:
                    T = shift @ T @ np.linalg.inv(shift)
                    r = emul.linearize(idx, S, T[:3, :3], T[:3, 3], radius=radius, wd=1, stats=True)
                    out[sweep].append((r, S.state.copy()))
            for (a, sa), (b, sb) in zip(out[True], out[False]):
                assert a["n_eff"] == b["n_eff"] and a["n_pt"] == b["n_pt"] and a["searched"] == b["searched"], name
                assert np.array_equal(a["H_upper"], b["H_upper"]) and np.array_equal(a["g"], b["g"]) and a["sum_r2"] == b["sum_r2"], name
                assert np.array_equal(sa, sb), name
                swept += int((b["stats"][:, 1] > 1).sum())           # queries whose ring walk went beyond the centre block
    assert swept > 10_000


def test_certificate_torture():
    """The certificate must never outlive its set.  Histories that try to break it: a long drift in steps of every size (the charged
    moves accumulate; the float store of the query position jitters), steps that undo each other, a lattice with duplicated points (no
    gap between the 5th and 6th neighbour: never certified), a sparse cloud where most queries have no 5 neighbours inside the radius
    (OUT certificates), coordinates far from the origin.  After every step: per-point results == oracle, sums bitwise == full search."""
    rng = np.random.default_rng(77)
    g = np.arange(0, 10, dtype=np.float32) * 0.3
    lattice = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    lattice = np.concatenate([lattice, lattice[::7]])
    cases = {
        "cylinder": (h.scene_cylinder(6000, seed=5, noise=0.01), 1.0),
        "fixture": (h.cylinder_cloud(), 1.0),
        "lattice_dups": (lattice, 0.7),
        "far_corridor": ((h.scene_corridor(8000, seed=6, length=20.0).astype(np.float64) + np.array([3.0e4, -2.0e4, 500.0])).astype(np.float32), 0.8),
    }
    for name, (tgt, radius) in cases.items():
        src = (tgt[::2] + rng.normal(0, 0.004, tgt[::2].shape)).astype(np.float32)
        if name == "lattice_dups":
            src = (tgt[::3] + np.float32(0.11)).astype(np.float32)
        idx, tree = emul.Index(tgt, radius), po.KdTree(tgt)
        Sc, Sf = emul.Source(src), emul.Source(src)
        c = src.astype(np.float64).mean(axis=0)
        T = np.eye(4)
        steps = [1e-7, 3e-7, 1e-6, 1e-5, -1e-5, 1e-4, 3e-4, -3e-4, 1e-3, 2e-3, 1e-6, 1e-6, 5e-3, 1e-2, 3e-2, -3e-2, 1e-6, 0.1, 1e-5, 1e-5]
        searched = []
        for k, sz in enumerate(steps):
            ang = sz / max(np.linalg.norm(src.astype(np.float64) - c, axis=1).max(), 1.0)
            dT = h.pose6d_matrix(sz * 0.6, -sz * 0.3, sz * 0.2, ang * 0.5, -ang * 0.3, ang)
            shift = np.eye(4); shift[:3, 3] = c
            T = shift @ dT @ np.linalg.inv(shift) @ T                # rotate about the cloud centre: rotation and translation partly cancel
            x = emul.linearize(idx, Sc, T[:3, :3], T[:3, 3], radius=radius, wd=1, plan="cert", debug=True)
            y = emul.linearize(idx, Sf, T[:3, :3], T[:3, 3], radius=radius, wd=1, plan="full")
            ref = po.linearize(tree, src, T[:3, :3], T[:3, 3], po.default_lin_params(radius, 1), debug=True)
            # (far from the origin the 5x3 systems [q_j] x = -1 are badly conditioned: the oracle's and the replay's plane fits differ
            # by rounding, amplified - the point here is the neighbour sets, flags and the bitwise equality with the full search)
            # (likewise the lattice: rank-deficient neighbourhoods, planes decided by the trailing pivots)
            loose = name in ("far_corridor", "lattice_dups")
            assert_same(x, ref, normals_atol=1e-4 if loose else 1e-8, h_rtol=1e-6 if loose else 1e-11)
            assert x["n_eff"] == y["n_eff"] and np.array_equal(x["H_upper"], y["H_upper"]) and np.array_equal(x["g"], y["g"]), (name, k)
            searched.append(x["searched"])
        assert searched[0] == len(src)                               # fresh state: everything is searched
        if name not in ("lattice_dups", "far_corridor"):       # (50 km from the origin a float ulp is 4 mm: the allowance for the float store
                                                               #  of the query position eats most certificates - correctly)
            assert min(searched[1:4]) < 0.05 * len(src), (name, searched)     # micrometre steps: (almost) nothing is
        assert max(searched[1:]) <= len(src)


def test_team_search_algorithm_keeps_the_sums_and_its_certificates_hold():
    """search.hpp team_search6 (what a sparse wave of k_lin runs for the queries whose warm bound lies inside their 27-cell block):
    the scalar replay of its algorithm - rows cut by team_row, every point inside the bound ranked by (distance bits, index), the
    seventh distance as the lower bound of the certificate - takes those queries instead of the lock-step search.  Walks that mix
    micrometre steps with steps of several neighbour spacings, on scenes with ties, duplicates, OUT points and coordinates far from
    the origin: after every step per-point results == oracle and sums bitwise == the all-search path; the certificates the team
    writes are used by the later steps, so a lower bound that claimed too much would surface there."""
    rng = np.random.default_rng(99)
    g = np.arange(0, 10, dtype=np.float32) * 0.3
    lattice = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    lattice = np.concatenate([lattice, lattice[::7]])
    cases = {
        "cylinder": (h.scene_cylinder(6000, seed=15, noise=0.01), 1.0),
        "fixture": (h.cylinder_cloud(), 1.0),
        "lattice_dups": (lattice, 0.7),
        "far_corridor": ((h.scene_corridor(8000, seed=16, length=20.0).astype(np.float64) + np.array([3.0e4, -2.0e4, 500.0])).astype(np.float32), 0.8),
        "planes": (h.scene_planes(9000, seed=3), 0.6),
    }
    served_total = 0
    for name, (tgt, radius) in cases.items():
        src = (tgt[::2] + rng.normal(0, 0.004, tgt[::2].shape)).astype(np.float32)
        if name == "lattice_dups":
            src = (tgt[::3] + np.float32(0.11)).astype(np.float32)
        idx, tree = emul.Index(tgt, radius), po.KdTree(tgt)
        idx.set_team(True)
        St, Sf = emul.Source(src), emul.Source(src)
        c = src.astype(np.float64).mean(axis=0)
        T = np.eye(4)
        steps = [1e-6, 1e-4, 1e-3, 3e-3, -3e-3, 1e-2, 2e-2, 1e-5, 4e-2, -4e-2, 1e-3, 8e-2, 1e-6, 5e-3, 0.15, 1e-4, 1e-3]
        for k, sz in enumerate(steps):
            ang = sz / max(np.linalg.norm(src.astype(np.float64) - c, axis=1).max(), 1.0)
            dT = h.pose6d_matrix(sz * 0.6, -sz * 0.3, sz * 0.2, ang * 0.5, -ang * 0.3, ang)
            shift = np.eye(4); shift[:3, 3] = c
            T = shift @ dT @ np.linalg.inv(shift) @ T
            x = emul.linearize(idx, St, T[:3, :3], T[:3, 3], radius=radius, wd=1, plan="cert", debug=True)
            idx.set_team(False)
            y = emul.linearize(idx, Sf, T[:3, :3], T[:3, 3], radius=radius, wd=1, plan="full")
            idx.set_team(True)
            ref = po.linearize(tree, src, T[:3, :3], T[:3, 3], po.default_lin_params(radius, 1), debug=True)
            loose = name in ("far_corridor", "lattice_dups")
            assert_same(x, ref, normals_atol=1e-4 if loose else 1e-8, h_rtol=1e-6 if loose else 1e-11)
            assert x["n_eff"] == y["n_eff"] and np.array_equal(x["H_upper"], y["H_upper"]) and np.array_equal(x["g"], y["g"]), (name, k)
        served_total += idx.team_served()
    assert served_total > 20_000                 # the team did take most of the searches of these walks


def test_team_rows_cover_the_ball():
    """team_row (the cell-table phase of one row, run-time offsets): every target point closer than the bound lies in one of the nine
    intervals whenever team_bound calls the query tight - checked through the full replay above; here the complementary property:
    with the team on, a "cert" walk searches exactly the points the walk without it searches (certificates are as strong)."""
    tgt = h.scene_cylinder(5000, seed=21, noise=0.01)
    src = (tgt[::2] + np.random.default_rng(5).normal(0, 0.004, tgt[::2].shape)).astype(np.float32)
    idx = emul.Index(tgt, 1.0)
    Sa, Sb = emul.Source(src), emul.Source(src)
    T = np.eye(4)
    tot_a = tot_b = 0
    for k, sz in enumerate([1e-5, 1e-3, 5e-3, 1e-3, 1e-2, 1e-4, 2e-3]):
        T = h.pose6d_matrix(sz, -sz * 0.5, sz * 0.25, 0.0, 0.0, sz * 0.1) @ T
        idx.set_team(True)
        a = emul.linearize(idx, Sa, T[:3, :3], T[:3, 3], wd=1, plan="cert")
        idx.set_team(False)
        b = emul.linearize(idx, Sb, T[:3, :3], T[:3, 3], wd=1, plan="cert")
        assert np.array_equal(a["H_upper"], b["H_upper"]) and a["n_eff"] == b["n_eff"]
        tot_a += a["searched"]; tot_b += b["searched"]
    # the team's seventh-neighbour bound is exact, the lock-step search's a lower bound of it: the team never searches more
    assert tot_a <= tot_b, (tot_a, tot_b)


def test_knn_with_ties_duplicates_and_outside_queries():
    g = np.arange(0, 12, dtype=np.float32) * 0.25
    tgt = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    tgt = np.concatenate([tgt, tgt[::9]])                                     # exact duplicates on top of the lattice ties
    rng = np.random.default_rng(5)
    q = np.concatenate([tgt[:300] + 0.125, rng.uniform(-3, 6, (1500, 3)), [[100, 100, 100]], tgt[:50]]).astype(np.float32)
    tree = po.KdTree(tgt)
    for cell, sx in ((0.0, 4), (0.3, 4), (0.3, 1), (0.0, 2), (0.3, 16)):          # auto cell / cells of ~one lattice step; x sub-cells
        idx = emul.Index(tgt, 1.0, cell=cell, x_subdiv=sx)
        for k in (1, 5):
            gi, gd = emul.knn(idx, q, k=k)
            oi, od = tree.knn(q, k=k)
            assert np.array_equal(gi, oi) and np.array_equal(gd.view(np.uint32), od.view(np.uint32))
        gi, gd = emul.knn(idx, q, k=5, max_radius=0.6)
        oi, od = tree.knn(q, k=5)
        inside = od <= np.float32(0.36)
        assert np.array_equal(gi[inside], oi[inside]) and np.all(gi[~inside] == -1)


def test_empty_space_between_clusters():
    """Queries in the void between two clusters, radius 3 m over ~0.25 m cells: a dozen rings, faces culled by the field."""
    rng = np.random.default_rng(8)
    a = rng.uniform(0, 2, (4000, 3)); b = rng.uniform(0, 2, (4000, 3)) + np.array([9.0, 0.5, 0.0])
    tgt = np.concatenate([a, b]).astype(np.float32)
    q = np.concatenate([rng.uniform(2, 9, (1500, 3)) * np.array([1, 0.3, 0.3]), rng.uniform(-4, 14, (500, 3)), tgt[:100] + 0.01]).astype(np.float32)
    oi, od = po.KdTree(tgt).knn(q, k=5)
    for gap, sx in ((True, 4), (False, 4), (True, 1), (True, 8)):
        idx = emul.Index(tgt, 3.0, gap_field=gap, x_subdiv=sx)
        assert idx.cell < 1.0 and (idx.gap_cap >= 2) == gap
        gi, gd = emul.knn(idx, q, k=5)
        assert np.array_equal(gi, oi) and np.array_equal(gd.view(np.uint32), od.view(np.uint32))
        bi, bd = emul.knn(idx, q, k=5, max_radius=3.0)
        inside = od < np.float32(9.0)
        assert np.array_equal(bi[inside], oi[inside])


def test_reduced_instruction_plane_fit_is_the_same_factorisation():
    """plane_fit_qr_fast vs plane_fit_qr (Eigen-shaped) vs the oracle on random and degenerate neighbourhoods."""
    rng = np.random.default_rng(3)
    worst = 0.0
    for trial in range(3000):
        c = rng.uniform(-50, 50, 3)
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        u = np.cross(n, [1, 0, 0.3]); u /= np.linalg.norm(u); v = np.cross(n, u)
        Q = c + np.outer(rng.uniform(-0.4, 0.4, 5), u) + np.outer(rng.uniform(-0.4, 0.4, 5), v) + np.outer(rng.normal(0, 0.01, 5), n)
        kind = trial % 6
        if kind == 3:
            Q[:, 2] = 0.0                                   # rank 2: a coordinate that is exactly zero (floor points of the fixture)
        if kind == 4:
            Q = Q.astype(np.float32).astype(np.float64)     # what the kernel sees: float coordinates
        if kind == 5:
            Q[:, 0] = 0.0; Q[:, 1] *= 1e-3                  # nearly collinear, far from well conditioned
        a, b = emul.plane_fit(Q, False), emul.plane_fit(Q, True)
        scale = max(np.linalg.norm(a), 1e-300)
        cond = np.linalg.cond(Q[:, np.abs(Q).max(0) > 0]) if kind != 5 else 1e8
        assert np.linalg.norm(a - b) <= 1e-13 * cond * scale + 1e-300, (trial, a, b)
        worst = max(worst, np.linalg.norm(a - b) / scale)
    assert worst < 1e-6


def test_ring_walk_cost_counters():
    """The visit counters the host replay exposes (candidates, rows, table loads, 4-candidate trips; tests/emul.py wave_cost) on a
    corridor misaligned by many cells - what C4's first iterations are.  They are the input of the design notes in
    profiles/r02_ablation.md; here only their consistency is pinned."""
    tgt = h.scene_corridor(150_000, seed=100, length=30.0)
    rng = np.random.default_rng(1100)
    src = (tgt + rng.normal(0, 0.01, tgt.shape)).astype(np.float32)
    idx, S = emul.Index(tgt, 1.0), emul.Source(src)
    idx.set_sweep(False)                                                 # the ring walk (dcreg_knn; k_lin sweeps rows instead)
    emul.linearize(idx, S, np.eye(3), np.zeros(3), wd=1)
    T = h.pose6d_matrix(0.05, -0.08, 0.03, h.deg2rad(0.2), h.deg2rad(-0.1), h.deg2rad(3.0))   # 0.8 m at the ends of 30 m
    out = emul.linearize(idx, S, T[:3, :3], T[:3, 3], wd=1, stats=True, trace_cap=512)
    st = out["stats"].astype(np.int64)
    assert out["n_eff"] > 100_000
    assert st[:, 0].mean() > 40 and (st[:, 1] > 1).mean() > 0.2          # the case does walk rings
    assert np.all(st[:, 5] * 4 >= st[:, 0])                              # every candidate sits in a 4-wide trip
    used = out["trace"][:, -1].astype(np.int64) // 2                      # ring rows / end cells whose cell table was looked up
    assert np.array_equal(np.minimum(used, 255), np.minimum(st[:, 6], 255)) and used.sum() > 0
    # neither the empty-space field nor the warm bound changes a result: same sums, bitwise, without them
    S2 = emul.Source(src)
    out2 = emul.linearize(emul.Index(tgt, 1.0, gap_field=False), S2, T[:3, :3], T[:3, 3], wd=1, warm=False)
    assert out2["n_eff"] == out["n_eff"] and np.array_equal(out2["H_upper"], out["H_upper"]) and np.array_equal(out2["g"], out["g"])


def test_far_from_the_origin_and_very_dense_cells():
    """Coordinates ~1e5 m from the origin (float spacing 8 mm: heavy quantisation, many exact ties, cell arithmetic in double) and a
    cloud whose 60 k points sit in a 2 cm cube (a few cells hold tens of thousands of points: runs far longer than any trimmed
    x-interval): exact k-NN against the oracle, bounded and unbounded."""
    rng = np.random.default_rng(21)
    base = h.scene_corridor(20000, seed=3)[:, :3] if hasattr(h, "scene_corridor") else rng.uniform(0, 30, (20000, 3)).astype(np.float32)
    far = (base.astype(np.float64) + np.array([1.0e5, -2.0e5, 3.0e4])).astype(np.float32)
    q = np.concatenate([far[::7] + rng.normal(0, 0.05, far[::7].shape).astype(np.float32), far[:200]]).astype(np.float32)
    oi, od = po.KdTree(far).knn(q, k=5)
    for sx in (1, 8):
        idx = emul.Index(far, 1.0, x_subdiv=sx)
        gi, gd = emul.knn(idx, q, k=5)
        assert np.array_equal(gi, oi) and np.array_equal(gd.view(np.uint32), od.view(np.uint32))
    dense = (rng.uniform(0, 0.02, (60000, 3)) + np.array([5.0, 5.0, 5.0])).astype(np.float32)
    qd = np.concatenate([dense[::300] + 0.001, rng.uniform(4.5, 5.5, (300, 3)), [[5.0, 5.0, 5.0]]]).astype(np.float32)
    oi, od = po.KdTree(dense).knn(qd, k=5)
    idx = emul.Index(dense, 1.0)
    gi, gd = emul.knn(idx, qd, k=5)
    assert np.array_equal(gi, oi) and np.array_equal(gd.view(np.uint32), od.view(np.uint32))
    bi, bd = emul.knn(idx, qd, k=5, max_radius=0.3)
    inside = od < np.float32(0.09)
    assert np.array_equal(bi[inside], oi[inside])


def test_the_rows_of_a_ball_hold_every_point_inside_it():
    """The small-frame advance pass (kernels.hpp k_advance_team) lists the (y,z) rows of a query's ball - all rows of its bounding square,
    or the rows whose occupancy bit is set in the words of its z layers - and cuts each to the ball (search.hpp ball_cells / ball_row).
    Replayed on the host for random queries and bounds from a fraction of a cell to a dozen cells, inside, at the border of and far
    outside the cloud: the points it finds below the bound are EXACTLY the brute-force ones (float distances as the device forms them),
    ranked by (distance, index)."""
    rng = np.random.default_rng(17)
    scenes = [(h.scene_corridor(30_000, seed=2, length=30.0), 1.0, 0.0), (h.scene_planes(20_000, seed=3), 0.5, 0.0),
              (h.cylinder_cloud(), 1.0, 0.0), (h.scene_cylinder(20_000, seed=4, noise=0.01), 2.0, 0.12)]
    served = left = 0
    for tgt, radius, cell in scenes:
        idx = emul.Index(tgt, radius, cell=cell)
        t32 = tgt.astype(np.float32)
        for case in range(300):
            base = t32[rng.integers(0, len(t32))]
            q = (base + rng.normal(0, float(rng.choice([0.0, 0.01, 0.3, 3.0])), 3)).astype(np.float32)
            if case % 50 == 0:
                q = (t32.max(0) + np.float32(rng.uniform(0.0, 40.0))).astype(np.float32)          # beyond a corner of the grid
            r = float(rng.choice([0.05, 0.3, 1.0])) * radius * 1.05
            bound = np.float32(r * r)
            n, got_idx, got_d2 = emul.ball_query(idx, q, bound)
            if n < 0:
                left += 1
                continue
            served += 1
            d = t32 - q                                                   # float32, un-fused, (x^2 + y^2) + z^2: dist2_nofma
            d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
            inside = np.flatnonzero(d2 < bound)
            assert n == len(inside), (case, n, len(inside), q, r)
            order = inside[np.lexsort((inside, d2[inside].view(np.uint32)))][:7]
            assert np.array_equal(got_idx[:len(order)], order.astype(np.int32)) and np.all(got_idx[len(order):] == -1), (case, got_idx, order)
            assert np.array_equal(got_d2[:len(order)].view(np.uint32), d2[order].view(np.uint32))
    assert served > 900 and left < 200, (served, left)
