"""Second engine of the reference (TestRunner::Point2PlaneICP, icp_test_runner.cpp:2064-2830: Pose6D state, LOAM
Jacobian).  No committed trace of the reference exercises it, so the oracle is pinned on mathematics here (the LOAM
row is the derivative of the Z-Y-X Euler rotation; at zero angles it coincides with the SO(3) row, whose traces ARE
pinned) and the HIP path is compared with the oracle on the GPU."""
import numpy as np
import pytest

import helpers as h
from oracle import pyoracle as po


def test_euler_rotation_derivatives_match_finite_differences():
    rng = np.random.default_rng(0)
    for _ in range(5):
        rpy = rng.uniform(-1.2, 1.2, 3)
        dR = po.euler_dR(*rpy)
        for k in range(3):
            e = np.zeros(3); e[k] = 1e-6
            Rp = h.pose6d_matrix(0, 0, 0, *(rpy + e))[:3, :3]
            Rm = h.pose6d_matrix(0, 0, 0, *(rpy - e))[:3, :3]
            assert np.allclose(dR[k], (Rp - Rm) / 2e-6, atol=1e-9)


def test_euler_row_equals_so3_row_at_zero_angles():
    """R = I: d(R p)/d(roll,pitch,yaw) = [e_k] x p, the right-perturbation Jacobian of SO(3) at the identity."""
    pts = h.cylinder_cloud()
    tree = po.KdTree(pts)
    T = h.pose6d_matrix(0.01, 0.01, 0.01, 0.0, 0.0, 0.0)
    so3 = po.linearize(tree, pts, T[:3, :3], T[:3, 3], po.default_lin_params(1.0, 0))
    eul = po.linearize(tree, pts, T[:3, :3], T[:3, 3], po.default_lin_params(1.0, 0, euler_rpy=(0.0, 0.0, 0.0)))
    assert eul["n_eff"] == so3["n_eff"] == 871
    assert h.rel_err(eul["H_upper"], so3["H_upper"]) < 1e-12 and h.rel_err(eul["g"], so3["g"]) < 1e-10


def test_euler_gradient_is_the_derivative_of_the_frozen_objective():
    """g = A^T b must be minus the derivative of 0.5 * sum (c . (R(rpy) p + t) + s d)^2 with the correspondences, normals
    and weights frozen - checked through the directional derivative of sum_b2 along each translation axis, where freezing
    is exact for an infinitesimal step (translation columns of A are c itself)."""
    pts = h.scene_cylinder(20_000, seed=3, noise=0.01)
    tree = po.KdTree(pts)
    rpy = (0.02, -0.03, 0.05)
    T = h.pose6d_matrix(0.05, -0.02, 0.03, *rpy)
    out = po.linearize(tree, pts[::2], T[:3, :3], T[:3, 3], po.default_lin_params(1.0, 0, euler_rpy=rpy), debug=True)
    ok = out["flag"] == 1
    n, r, s = out["normal"][ok], out["r"][ok], out["s"][ok]
    c = (s[:, None] * n).astype(np.float32).astype(np.float64)
    b = -(s * r).astype(np.float32).astype(np.float64)
    assert np.allclose(out["g"][3:], c.T @ b, rtol=1e-9, atol=1e-9)
    dR = po.euler_dR(*rpy)
    P = pts[::2][ok].astype(np.float64)
    A_rot = np.stack([np.einsum("ij,ij->i", c, P @ dR[k].T) for k in range(3)], 1)
    assert np.allclose(out["g"][:3], A_rot.T @ b, rtol=1e-9, atol=1e-8)
    A = np.concatenate([A_rot, c], 1)
    Hn = A.T @ A
    iu = np.triu_indices(6)
    assert np.allclose(np.asarray(out["H_upper"]), Hn[iu], rtol=1e-9, atol=1e-6)


def test_euler_engine_converges_on_the_fixture():
    pts = h.cylinder_cloud()
    tree = po.KdTree(pts)
    cfg = po.default_config(search_radius=1.0, max_iterations=30, kappa_target=10.0, std_reg_gamma=100.0)
    res, logs, pose = po.icp_run_euler(tree, pts, (0.0, 0.0, 0.0, 0.01, 0.01, 0.01), "ME-SR", cfg)
    assert res.status == 0 and res.converged == 1 and 2 <= res.iterations <= 30
    assert logs[0].n_eff == 871                                   # same correspondences as the SO(3) engine's first iteration
    T = np.array(logs[-1].T[:]).reshape(4, 4)
    assert np.allclose(T, h.pose6d_matrix(*pose[3:], *pose[:3]), atol=1e-12)
    assert logs[-1].trans_err < 0.03 and logs[-1].rot_err_deg < 0.1
    # additive Euler update: pose_k+1 = pose_k + dx_k
    acc = np.array([0.0, 0.0, 0.0, 0.01, 0.01, 0.01])
    for L in logs:
        acc = acc + np.array(L.dx[:])
    assert np.allclose(acc, pose, atol=1e-14)


@pytest.mark.gpu
def test_euler_hip_matches_oracle():
    from dcreg_amd import api
    ctx = api.Context(0)
    try:
        tgt = h.scene_cylinder(50_000, seed=5, noise=0.01)
        src = tgt[::3].copy()
        tree = po.KdTree(tgt)
        ctx.set_target(tgt, 1.0); ctx.set_source(src)
        rpy = (0.004, -0.003, 0.009)
        T = h.pose6d_matrix(0.05, -0.08, 0.03, *rpy)
        gpu = ctx.linearize(T[:3, :3], T[:3, 3], api.default_lin_params(1.0, 0, euler_rpy=rpy))
        ref = po.linearize(tree, src, T[:3, :3], T[:3, 3], po.default_lin_params(1.0, 0, euler_rpy=rpy))
        assert gpu["n_eff"] == ref["n_eff"] > 1000 and gpu["n_pt"] == ref["n_pt"]
        assert h.rel_err(gpu["H_upper"], ref["H_upper"]) < 1e-9 and h.rel_err(gpu["g"], ref["g"]) < 1e-8
        so3 = ctx.linearize(T[:3, :3], T[:3, 3], api.default_lin_params(1.0, 0))
        assert so3["n_eff"] == gpu["n_eff"] and h.rel_err(so3["H_upper"], gpu["H_upper"]) > 1e-6    # a different row
        # the engine: same iterations, updates and final pose as the oracle's restatement
        pts = h.cylinder_cloud()
        ctx.set_target(pts, 1.0); ctx.set_source(pts)
        cfg = api.default_config(search_radius=1.0, max_iterations=30, KAPPA_TARGET=10.0, STD_REG_GAMMA=100.0)
        ocfg = po.default_config(search_radius=1.0, max_iterations=30, kappa_target=10.0, std_reg_gamma=100.0)
        for method in ("ME-SR", "Ours"):
            p0 = (0.0, 0.0, 0.0, 0.01, 0.01, 0.01)
            res, logs, pose = ctx.icp_run_euler(p0, method, cfg)
            ores, ologs, opose = po.icp_run_euler(po.KdTree(pts), pts, p0, method, ocfg)
            assert (res.converged, res.iterations, res.status) == (ores.converged, ores.iterations, ores.status)
            assert len(logs) == len(ologs)
            for a, b in zip(logs, ologs):
                assert a.effective_points == b.n_eff
                assert np.allclose(a.update_dx[:], b.dx[:], rtol=0, atol=2e-7)
            assert np.allclose(pose, opose, atol=1e-6)
            cov = np.array(res.icp_cov[:]).reshape(6, 6)
            assert np.allclose(cov, cov.T, atol=1e-12 * np.abs(cov).max()) and np.linalg.eigvalsh(cov).min() >= 1e-9 * 0.999
    finally:
        ctx.close()


@pytest.mark.gpu
def test_runner_selects_the_euler_engine(tmp_path):
    """icp.use_so3_parameterization: false (Config field utils.hpp:170) routes the named methods through the second engine."""
    import os, subprocess
    runner = os.path.join(h.REPO, "dcreg_amd", "bin", "icp_test_runner")
    cfg = open(os.path.join(h.REPO, "configs", "icp.yaml")).read().replace("normal_nn: 5", "normal_nn: 5\n  use_so3_parameterization: false")
    assert "use_so3_parameterization" in cfg
    ypath = os.path.join(str(tmp_path), "icp_euler.yaml")
    open(ypath, "w").write(cfg)
    out = str(tmp_path) + "/"
    p = subprocess.run([runner, ypath, out], cwd=h.REPO, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "USE_SO3 ICP: 0" in p.stdout
    rows = h.read_csv_rows(out + "all_results.csv")
    assert [r["Method"] for r in rows] == ["FCN-SR", "ME-SR", "ME-TReg", "ME-TSVD"]
    pts = h.cylinder_cloud()
    ocfg = po.default_config(search_radius=1.0, max_iterations=30, kappa_target=10.0, std_reg_gamma=100.0)
    ores, ologs, _ = po.icp_run_euler(po.KdTree(pts), pts, (0.0, 0.0, 0.0, 0.01, 0.01, 0.01), "ME-SR", ocfg)
    me = [r for r in rows if r["Method"] == "ME-SR"][0]
    assert int(me["Iterations"]) == ores.iterations and int(me["Converged"]) == ores.converged
    assert abs(float(me["Trans_Error_m"]) - ologs[-1].trans_err) < 1e-6
