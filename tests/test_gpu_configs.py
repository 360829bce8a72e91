"""BASELINE.json configs at their stated sizes, engine level: FULL multi-iteration ICP runs through the HIP path
(dcreg_icp_run / icp_test_runner, i.e. through the C-ABI) against the CPU oracle run on the same seeded inputs, compared on
every iteration -- the reference loop these follow is DCReg/src/icp_test_runner.cpp:1694-2004.

  C2  100 k x 100 k cylinder pair, 20 iterations            (configs[1])
  C3  PK01-like parking lot: 200 k map, 8 k frame and the 200 k-frame variant, R = 0.5, Schur + PCG, icp_pk01.yaml (configs[2])
  C4  1 M x 1 M corridor, 50 iterations                       (configs[3])
  +   the covariance of the SO(3) engine (:2014-2037) against the oracle's, on converged runs

Tolerances: counts and masks exact; H <= 1e-7 relative (north_star: 1e-5); updates and poses <= 1e-8 absolute."""
import os
import subprocess

import numpy as np
import pytest

import helpers as h
from dcreg_amd import api
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

T_SMALL = h.pose6d_matrix(0.05, -0.08, 0.03, h.deg2rad(0.2), h.deg2rad(-0.1), h.deg2rad(0.5))   # bench.py's initial misalignment


def noisy_pair(gen, seed):
    tgt = gen()
    rng = np.random.default_rng(seed + 1000)
    return tgt, (tgt + rng.normal(0, 0.01, tgt.shape)).astype(np.float32)


def cfg_pair(radius, iters, wd, thresh_rot=0.0, thresh_trans=0.0, gt=None):
    kw = dict(search_radius=radius, max_iterations=iters, KAPPA_TARGET=10.0, STD_REG_GAMMA=100.0, CONVERGENCE_THRESH_ROT=thresh_rot,
              CONVERGENCE_THRESH_TRANS=thresh_trans, use_weight_derivative=wd, always_compute_schur=1)
    okw = dict(search_radius=radius, max_iterations=iters, kappa_target=10.0, std_reg_gamma=100.0, thresh_rot=thresh_rot,
               thresh_trans=thresh_trans, use_weight_derivative=wd, always_compute_schur=1, num_threads=8)
    if gt is not None:
        kw["gt_matrix"] = gt
        okw["gt"] = gt
    return api.default_config(**kw), po.default_config(**okw)


def assert_runs_equal(res, logs, ores, ologs, h_rtol=1e-7, dx_atol=1e-8, pose_atol=1e-8):
    assert res.iterations == ores.iterations and res.converged == ores.converged and res.status == ores.status
    assert len(logs) == len(ologs) > 0
    for L, O in zip(logs, ologs):
        assert L.effective_points == O.n_eff and L.corr_pt_count == O.n_pt, (L.iter_count, L.effective_points, O.n_eff)
        assert list(L.analysis.degenerate_mask[:]) == list(O.an.mask[:]), L.iter_count
        assert h.rel_err(L.H_upper[:], O.H_upper[:]) < h_rtol, (L.iter_count, h.rel_err(L.H_upper[:], O.H_upper[:]))
        if L.iter_count == 0:
            # same pose on both sides: pure kernel parity.  g = sum_i a_i b_i cancels heavily; its rounding scale is the
            # Cauchy-Schwarz bound sqrt(H_jj * sum b^2).  (Later iterations: the trajectories differ by ~1e-12 in the pose, which
            # moves g by H * that -- up to 1e-3 at 1 M points -- while g itself goes to zero; the update and the pose are compared instead.)
            Hd = np.diag(api.unpack_hessian(np.array(O.H_upper[:])))
            assert np.all(np.abs(np.array(L.gradient[:]) - O.gradient[:]) <= 1e-9 * np.sqrt(Hd * 2.0 * O.objective) + 1e-12)
        assert np.allclose(L.update_dx[:], O.dx[:], rtol=0, atol=dx_atol), (L.iter_count, np.max(np.abs(np.array(L.update_dx[:]) - O.dx[:])))
        assert np.isclose(L.rmse, O.rmse, rtol=1e-9) and np.isclose(L.fitness, O.fitness, rtol=1e-12)
        for a, b in ((L.analysis.cond_schur_rot, O.an.cond_schur_rot), (L.analysis.cond_schur_trans, O.an.cond_schur_trans)):
            assert np.isclose(a, b, rtol=1e-6)
        assert np.allclose(L.transform_matrix[:], O.T[:], rtol=0, atol=pose_atol)
    # final SE(3) pose: north_star asks for 1e-5
    assert np.allclose(res.R[:], ores.R[:], rtol=0, atol=pose_atol) and np.allclose(res.t[:], ores.t[:], rtol=0, atol=pose_atol)


def assert_cov_equal(res, ores):
    a, b = np.array(res.icp_cov[:]).reshape(6, 6), np.array(ores.cov[:]).reshape(6, 6)
    assert np.allclose(a, a.T, rtol=0, atol=1e-12 * np.max(np.abs(a)))
    assert h.rel_err(a, b) < 1e-6, h.rel_err(a, b)          # inverse of H: its conditioning (1e3-1e5) amplifies the 1e-10 of H
    if res.converged:
        assert np.all(np.linalg.eigvalsh(0.5 * (a + a.T)) > 0)
    else:
        assert np.array_equal(a, 1e6 * np.eye(6))


@pytest.fixture(scope="module")
def ctx():
    c = api.Context(0)
    yield c
    c.close()


def test_c2_cylinder_100k_20_iterations(ctx):
    tgt, src = noisy_pair(lambda: h.scene_cylinder(100_000, seed=100, noise=0.01), 100)
    ctx.set_target(tgt, 1.0)
    ctx.set_source(src)
    cfg, ocfg = cfg_pair(1.0, 20, 1)
    res, logs = ctx.icp_run(T_SMALL, "Ours", cfg)
    ores, ologs = po.icp_run(po.KdTree(tgt), src, T_SMALL, "Ours", ocfg)
    assert res.iterations == 20 and logs[-1].effective_points > 99_000
    assert_runs_equal(res, logs, ores, ologs)
    assert_cov_equal(res, ores)                                  # fixed-length run: not converged -> 1e6 I on both sides


@pytest.mark.parametrize("method", ["Ours", "ME-TReg", "FCN-SR"])
def test_c3_pk01_standin_engine(ctx, method):
    """8 k-point frame against the 200 k-point prior map, the yaml's own poses, R = 0.5, thresholds 1e-3 / 1e-5."""
    tgt, src = h.scene_parkinglot()
    gt, T0 = h.pose6d_matrix(**h.PK01_GT), h.pose6d_matrix(**h.PK01_INIT)
    ctx.set_target(tgt, 0.5)
    ctx.set_source(src)
    cfg, ocfg = cfg_pair(0.5, 30, 0, 1e-5, 1e-3, gt.reshape(16))
    res, logs = ctx.icp_run(T0, method, cfg)
    ores, ologs = po.icp_run(po.KdTree(tgt), src, T0, method, ocfg)
    assert_runs_equal(res, logs, ores, ologs)
    assert_cov_equal(res, ores)
    if method == "Ours":
        assert res.converged == 1 and logs[0].analysis.degenerate_mask[3] == 1      # translation Schur block flagged at the start
        assert all(L.analysis.pcg_iterations == O.an.pcg_iterations > 0 for L, O in zip(logs, ologs) if O.an.is_degenerate)
        assert logs[-1].trans_error_vs_gt < 0.01 and np.isclose(logs[-1].trans_error_vs_gt, ologs[-1].trans_err, atol=1e-8)


def test_c3_pk01_standin_200k_frame(ctx):
    """Throughput variant: the whole map as one 200 k-point frame (sigma = 2 cm)."""
    tgt, src = h.scene_parkinglot(n_frame=200_000, frame_range=100.0)
    gt, T0 = h.pose6d_matrix(**h.PK01_GT), h.pose6d_matrix(**h.PK01_INIT)
    assert len(src) == 200_000
    ctx.set_target(tgt, 0.5)
    ctx.set_source(src)
    cfg, ocfg = cfg_pair(0.5, 30, 0, 1e-5, 1e-3, gt.reshape(16))
    res, logs = ctx.icp_run(T0, "Ours", cfg)
    ores, ologs = po.icp_run(po.KdTree(tgt), src, T0, "Ours", ocfg)
    assert logs[1].effective_points > 150_000
    assert_runs_equal(res, logs, ores, ologs)
    assert_cov_equal(res, ores)


def test_c3_pk01_yaml_through_the_runner(tmp_path):
    """configs/icp_pk01.yaml = the reference's config/icp_pk01.yaml verbatim except the paths; icp_test_runner output files
    against the oracle run of the same pair."""
    import sys
    sys.path.insert(0, os.path.join(h.REPO, "scripts"))
    import make_pk01_standin as gen
    tgt, src = gen.main()
    out = str(tmp_path) + "/"
    runner = os.path.join(h.REPO, "dcreg_amd", "bin", "icp_test_runner")
    p = subprocess.run([runner, os.path.join(h.REPO, "configs", "icp_pk01.yaml"), out], cwd=h.REPO, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "Source: 8000 points, Target: 200000 points" in p.stdout and "skipped" in p.stdout    # O3D / XICP / SuperLoc reported, not run
    rows = h.read_csv_rows(out + "all_results.csv")
    assert [r["Method"] for r in rows] == ["FCN-SR", "ME-SR", "ME-TReg", "ME-TSVD", "Ours"]          # std::map order of the supported ones
    gt, T0 = h.pose6d_matrix(**h.PK01_GT), h.pose6d_matrix(**h.PK01_INIT)
    tree = po.KdTree(tgt)
    for r in rows:
        _, ocfg = cfg_pair(0.5, 30, 0, 1e-5, 1e-3, gt.reshape(16))
        ocfg.always_compute_schur = 0
        ores, ologs = po.icp_run(tree, src, T0, r["Method"], ocfg)
        assert int(r["Iterations"]) == ores.iterations and int(r["Converged"]) == ores.converged
        assert np.isclose(float(r["Trans_Error_m"]), ologs[-1].trans_err, rtol=1e-4, atol=1e-7)
        assert np.isclose(float(r["Rot_Error_deg"]), ologs[-1].rot_err_deg, rtol=1e-4, atol=1e-7)
        hist = [x for x in h.read_csv_rows(out + "iteration_history.csv") if x["Method"] == r["Method"]]
        assert [int(x["CorrNum"]) for x in hist] == [O.n_eff for O in ologs]
        det = [x for x in h.read_csv_rows(out + "iteration_details_with_dx.csv") if x["Method"] == r["Method"]]
        for d, O in zip(det, ologs):
            assert np.allclose([float(d[k]) for k in ("dx_wx", "dx_wy", "dx_wz", "dx_x", "dx_y", "dx_z")], O.dx[:], rtol=0, atol=2e-8)


@pytest.mark.timeout(600)
def test_c4_corridor_1m_50_iterations(ctx):
    tgt, src = noisy_pair(lambda: h.scene_corridor(1_000_000, seed=100), 100)
    ctx.set_target(tgt, 1.0)
    ctx.set_source(src)
    cfg, ocfg = cfg_pair(1.0, 50, 1)
    res, logs = ctx.icp_run(T_SMALL, "Ours", cfg)
    ores, ologs = po.icp_run(po.KdTree(tgt), src, T_SMALL, "Ours", ocfg)
    assert res.iterations == 50 and logs[-1].effective_points > 990_000
    assert_runs_equal(res, logs, ores, ologs)
    # the second run from the same pose starts with the warm-start state of the converged one: same trajectory, bitwise
    res2, logs2 = ctx.icp_run(T_SMALL, "Ours", cfg)
    assert np.array_equal(np.array(res.R[:]), np.array(res2.R[:])) and np.array_equal(np.array(res.t[:]), np.array(res2.t[:]))
    assert all(np.array_equal(np.array(a.H_upper[:]), np.array(b.H_upper[:])) for a, b in zip(logs, logs2))


def test_covariance_of_the_so3_engine_on_the_committed_run(ctx):
    """`Ours` on the fixture (paper run, converges in 10 iterations): icp_cov = FullPivLU(H_last)^-1, PSD-clamped (:2014-2037)."""
    pts = h.cylinder_cloud()
    ctx.set_target(pts, 1.0)
    ctx.set_source(pts)
    T0 = h.pose6d_matrix(**h.PAPER_INIT)
    cfg, ocfg = cfg_pair(1.0, 30, 1, 1e-5, 1e-3)
    for method in ("Ours", "ME-SR", "ME-TReg"):
        res, logs = ctx.icp_run(T0, method, cfg)
        ores, ologs = po.icp_run(po.KdTree(pts), pts, T0, method, ocfg)
        assert res.converged == ores.converged and (method != "Ours" or res.converged == 1)
        assert_cov_equal(res, ores)
        if res.converged:
            Hl = api.unpack_hessian(np.array(logs[-1].H_upper[:]))
            assert h.rel_err(np.array(res.icp_cov[:]).reshape(6, 6) @ Hl, np.eye(6)) < 1e-8
