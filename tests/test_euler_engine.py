"""Second engine of the reference (TestRunner::Point2PlaneICP, icp_test_runner.cpp:2064-2830: Pose6D state, LOAM-shaped
Jacobian).  No committed trace of the reference exercises it: the row stays PARITY-UNPINNED against reference outputs.  What can be
checked is checked here: the oracle's default row (parameterization 1) is the expression of icp_test_runner.cpp:2299-2346 term for
term - its LOAM brackets multiplied by coeff.z, coeff.x, coeff.y, a cyclic permutation of LOAM's coeff.x, coeff.y, coeff.z that
makes the rotation columns something other than the derivative of the residual -, the additive exact row (parameterization 2) is
that derivative, the two differ exactly by the permutation, and the HIP path is compared with the oracle on the GPU for both."""
import math

import numpy as np
import pytest

import helpers as h
from oracle import pyoracle as po


def reference_row_as_written(rpy, p, c):
    """icp_test_runner.cpp:2299-2346 typed out again in Python (IEEE doubles, one operation at a time like un-contracted x86 code):
    p = the body-frame float point (laserCloudEffective), c = the float-stored weighted normal (coeffSel)."""
    roll, pitch, yaw = (float(v) for v in rpy)
    srx, crx = math.sin(pitch), math.cos(pitch)
    sry, cry = math.sin(yaw), math.cos(yaw)
    srz, crz = math.sin(roll), math.cos(roll)
    px, py, pz = (float(np.float32(v)) for v in p)
    cx, cy, cz = (float(np.float32(v)) for v in c)
    ox, oy, oz = py, pz, px              # pointOri.x = .y, .y = .z, .z = .x
    kx, ky, kz = cy, cz, cx              # coeff.x = .y,  .y = .z, .z = .x
    crx_sry = crx * sry; crz_sry = crz * sry; srx_sry = srx * sry; srx_srz = srx * srz
    arx = ((crx_sry * srz * ox + crx * crz_sry * oy - srx_sry * oz) * kz +
           (-srx_srz * ox - crz * srx * oy - crx * oz) * kx +
           (crx * cry * srz * ox + crx * cry * crz * oy - cry * srx * oz) * ky)
    ary = (((cry * srx_srz - crz_sry) * ox + (sry * srz + cry * crz * srx) * oy + crx * cry * oz) * kz +
           ((-cry * crz - srx_sry * srz) * ox + (cry * srz - crz * srx_sry) * oy - crx_sry * oz) * ky)
    arz = (((crz * srx_sry - cry * srz) * ox + (-cry * crz - srx_sry * srz) * oy) * kz +
           (crx * crz * ox - crx * srz * oy) * kx +
           ((sry * srz + cry * crz * srx) * ox + (crz_sry - cry * srx_srz) * oy) * ky)
    return np.array([arz, arx, ary, kz, kx, ky])


def test_oracle_row_is_the_reference_expression_term_for_term():
    """Same products, same sums, same order: the rows agree bit for bit wherever the six sines / cosines do (gcc merges each sin / cos
    pair of the oracle into glibc's sincos, whose last bit differs from math.sin / math.cos for a few arguments: those rows differ by an ulp
    or two of their largest term and nothing else)."""
    rng = np.random.default_rng(11)
    exact = 0
    for _ in range(200):
        rpy = rng.uniform(-1.5, 1.5, 3)
        p = rng.uniform(-30, 30, 3).astype(np.float32)
        c = rng.uniform(-1, 1, 3).astype(np.float32)
        lit, _ = po.euler_rows(rpy, p, c)
        want = reference_row_as_written(rpy, p, c)
        exact += int(np.array_equal(lit.view(np.uint64), want.view(np.uint64)))
        assert np.abs(lit - want).max() <= 4 * np.finfo(np.float64).eps * np.abs(p).max() * np.abs(c).max() * 3, (lit, want)
    assert exact >= 180, exact


def test_reference_row_is_the_exact_row_with_the_normal_permuted():
    """The reference multiplies LOAM's brackets by coeff z, x, y instead of x, y, z: its rotation columns for the weighted normal
    (cx, cy, cz) are the exact derivative's for (cz, cx, cy) - and NOT the derivative for (cx, cy, cz)."""
    rng = np.random.default_rng(12)
    worst = 0.0
    for _ in range(200):
        rpy = rng.uniform(-1.5, 1.5, 3)
        p = rng.uniform(-30, 30, 3).astype(np.float32)
        c = rng.uniform(-1, 1, 3).astype(np.float32)
        lit, ex = po.euler_rows(rpy, p, c)
        _, ex_perm = po.euler_rows(rpy, p, c[[2, 0, 1]])
        scale = np.abs(ex_perm[:3]).max() + 1e-300
        assert np.abs(lit[:3] - ex_perm[:3]).max() <= 1e-13 * max(scale, 1.0)
        assert np.array_equal(lit[3:], c.astype(np.float64)) and np.array_equal(ex[3:], c.astype(np.float64))
        worst = max(worst, np.abs(lit[:3] - ex[:3]).max() / (np.abs(ex[:3]).max() + 1e-300))
    assert worst > 1.0            # a different row, by more than its own size at some poses


def test_exact_row_is_the_derivative_of_the_rotated_point():
    rng = np.random.default_rng(13)
    for _ in range(20):
        rpy = rng.uniform(-1.2, 1.2, 3)
        p = rng.uniform(-5, 5, 3).astype(np.float32)
        c = rng.uniform(-1, 1, 3).astype(np.float32)
        _, ex = po.euler_rows(rpy, p, c)
        for k in range(3):
            e = np.zeros(3); e[k] = 1e-6
            Rp = h.pose6d_matrix(0, 0, 0, *(rpy + e))[:3, :3]
            Rm = h.pose6d_matrix(0, 0, 0, *(rpy - e))[:3, :3]
            fd = c.astype(np.float64) @ ((Rp - Rm) / 2e-6) @ p.astype(np.float64)
            assert abs(ex[k] - fd) < 1e-7 * (1 + abs(fd))


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


def test_exact_euler_row_equals_so3_row_at_zero_angles():
    """R = I: d(R p)/d(roll,pitch,yaw) = [e_k] x p, the right-perturbation Jacobian of SO(3) at the identity - for the exact row; the
    reference's row shares the translation block and the correspondences, not the rotation block."""
    pts = h.cylinder_cloud()
    tree = po.KdTree(pts)
    T = h.pose6d_matrix(0.01, 0.01, 0.01, 0.0, 0.0, 0.0)
    so3 = po.linearize(tree, pts, T[:3, :3], T[:3, 3], po.default_lin_params(1.0, 0))
    eul = po.linearize(tree, pts, T[:3, :3], T[:3, 3], po.default_lin_params(1.0, 0, euler_rpy=(0.0, 0.0, 0.0), euler_exact=True))
    assert eul["n_eff"] == so3["n_eff"] == 871
    assert h.rel_err(eul["H_upper"], so3["H_upper"]) < 1e-12 and h.rel_err(eul["g"], so3["g"]) < 1e-10
    ref = po.linearize(tree, pts, T[:3, :3], T[:3, 3], po.default_lin_params(1.0, 0, euler_rpy=(0.0, 0.0, 0.0)))
    assert ref["n_eff"] == 871
    Hs, Hr = po.unpack_H(so3["H_upper"]), po.unpack_H(ref["H_upper"])
    assert h.rel_err(Hr[3:, 3:], Hs[3:, 3:]) < 1e-12 and h.rel_err(Hr[:3, :3], Hs[:3, :3]) > 1e-3


def test_euler_gradient_is_the_derivative_of_the_frozen_objective():
    """g = A^T b must be minus the derivative of 0.5 * sum (c . (R(rpy) p + t) + s d)^2 with the correspondences, normals
    and weights frozen - checked through the directional derivative of sum_b2 along each translation axis, where freezing
    is exact for an infinitesimal step (translation columns of A are c itself)."""
    pts = h.scene_cylinder(20_000, seed=3, noise=0.01)
    tree = po.KdTree(pts)
    rpy = (0.02, -0.03, 0.05)
    T = h.pose6d_matrix(0.05, -0.02, 0.03, *rpy)
    out = po.linearize(tree, pts[::2], T[:3, :3], T[:3, 3], po.default_lin_params(1.0, 0, euler_rpy=rpy, euler_exact=True), debug=True)
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


def test_euler_engine_on_the_fixture():
    """With the exact row the engine converges on the fixture in a handful of iterations; with the reference's row (the default: it is what
    icp_test_runner.cpp:2323-2335 computes) the rotation block of H and g is not the residual's derivative and the same run wanders for
    all 30 iterations - which is what the reference would do: its own loader never selects this engine (utils.hpp:170 is not read)."""
    pts = h.cylinder_cloud()
    tree = po.KdTree(pts)
    p0 = (0.0, 0.0, 0.0, 0.01, 0.01, 0.01)
    for exact in (1, 0):
        cfg = po.default_config(search_radius=1.0, max_iterations=30, kappa_target=10.0, std_reg_gamma=100.0, euler_exact_jacobian=exact)
        res, logs, pose = po.icp_run_euler(tree, pts, p0, "ME-SR", cfg)
        assert res.status == 0 and logs[0].n_eff == 871           # same correspondences as the SO(3) engine's first iteration
        if exact:
            assert res.converged == 1 and 2 <= res.iterations <= 10
            assert logs[-1].trans_err < 0.03 and logs[-1].rot_err_deg < 0.1
        else:
            assert res.converged == 0 and res.iterations == 30
            assert logs[-1].rot_err_deg > 0.2
        T = np.array(logs[-1].T[:]).reshape(4, 4)
        assert np.allclose(T, h.pose6d_matrix(*pose[3:], *pose[:3]), atol=1e-12)
        # additive Euler update: pose_k+1 = pose_k + dx_k
        acc = np.array(p0)
        for L in logs:
            acc = acc + np.array(L.dx[:])
        assert np.allclose(acc, pose, atol=1e-14)


@pytest.mark.gpu
def test_euler_hip_matches_oracle():
    """Both rows through the C-ABI against the oracle: DCREG_PARAM_EULER (the reference's, default of dcreg_icp_run_euler) and
    DCREG_PARAM_EULER_EXACT (dcreg_config::euler_exact_jacobian)."""
    from dcreg_amd import api
    ctx = api.Context(0)
    try:
        tgt = h.scene_cylinder(50_000, seed=5, noise=0.01)
        src = tgt[::3].copy()
        tree = po.KdTree(tgt)
        ctx.set_target(tgt, 1.0); ctx.set_source(src)
        sums = {}
        T = h.pose6d_matrix(0.05, -0.08, 0.03, 0.004, -0.003, 0.009)
        for rpy in ((0.004, -0.003, 0.009), (0.4, -0.7, 1.1)):
            for exact in (False, True):
                # (the second set of angles does not belong to T: the row only reads euler_rpy, and large angles tell the 27 coefficients apart)
                gpu = ctx.linearize(T[:3, :3], T[:3, 3], api.default_lin_params(1.0, 0, euler_rpy=rpy, euler_exact=exact))
                ref = po.linearize(tree, src, T[:3, :3], T[:3, 3], po.default_lin_params(1.0, 0, euler_rpy=rpy, euler_exact=exact))
                assert gpu["n_eff"] == ref["n_eff"] > 1000 and gpu["n_pt"] == ref["n_pt"]
                assert h.rel_err(gpu["H_upper"], ref["H_upper"]) < 1e-9 and h.rel_err(gpu["g"], ref["g"]) < 1e-8
                sums[(rpy, exact)] = gpu
            a, b = sums[(rpy, False)], sums[(rpy, True)]
            assert a["n_eff"] == b["n_eff"] and h.rel_err(a["H_upper"], b["H_upper"]) > 1e-3     # the permutation is not cosmetic
        rpy = (0.004, -0.003, 0.009)
        T = h.pose6d_matrix(0.05, -0.08, 0.03, *rpy)
        so3 = ctx.linearize(T[:3, :3], T[:3, 3], api.default_lin_params(1.0, 0))
        assert so3["n_eff"] == sums[(rpy, True)]["n_eff"] and h.rel_err(so3["H_upper"], sums[(rpy, True)]["H_upper"]) > 1e-6    # a different row
        # the engine: same iterations, updates and final pose as the oracle's restatement, for both rows
        pts = h.cylinder_cloud()
        ctx.set_target(pts, 1.0); ctx.set_source(pts)
        for exact in (0, 1):
            cfg = api.default_config(search_radius=1.0, max_iterations=30, KAPPA_TARGET=10.0, STD_REG_GAMMA=100.0, euler_exact_jacobian=exact)
            ocfg = po.default_config(search_radius=1.0, max_iterations=30, kappa_target=10.0, std_reg_gamma=100.0, euler_exact_jacobian=exact)
            for method in ("ME-SR", "Ours"):
                p0 = (0.0, 0.0, 0.0, 0.01, 0.01, 0.01)
                res, logs, pose = ctx.icp_run_euler(p0, method, cfg)
                ores, ologs, opose = po.icp_run_euler(po.KdTree(pts), pts, p0, method, ocfg)
                assert (res.converged, res.iterations, res.status) == (ores.converged, ores.iterations, ores.status)
                assert res.converged == exact
                assert len(logs) == len(ologs)
                # (the reference's row does not contract: 30 iterations of a wandering pose amplify rounding differences, so those runs are
                # compared over their first iterations)
                n_cmp = len(logs) if exact else 6
                for a, b in list(zip(logs, ologs))[:n_cmp]:
                    assert a.effective_points == b.n_eff
                    assert np.allclose(a.update_dx[:], b.dx[:], rtol=0, atol=2e-7)
                if exact:
                    assert np.allclose(pose, opose, atol=1e-6)
                cov = np.array(res.icp_cov[:]).reshape(6, 6)
                assert np.allclose(cov, cov.T, atol=1e-12 * np.abs(cov).max()) and np.linalg.eigvalsh(cov).min() >= 1e-9 * 0.999
    finally:
        ctx.close()


@pytest.mark.gpu
def test_runner_selects_the_euler_engine(tmp_path):
    """icp.use_so3_parameterization: false (Config field utils.hpp:170) routes the named methods through the second engine; the additive key
    icp.euler_exact_jacobian chooses between the reference's row (default) and the exact derivative."""
    import os, subprocess
    runner = os.path.join(h.REPO, "dcreg_amd", "bin", "icp_test_runner")
    base = open(os.path.join(h.REPO, "configs", "icp.yaml")).read()
    pts = h.cylinder_cloud()
    for exact in (1, 0):
        cfg = base.replace("normal_nn: 5", "normal_nn: 5\n  use_so3_parameterization: false\n  euler_exact_jacobian: %s" % ("true" if exact else "false"))
        assert "use_so3_parameterization" in cfg
        ypath = os.path.join(str(tmp_path), "icp_euler_%d.yaml" % exact)
        open(ypath, "w").write(cfg)
        out = os.path.join(str(tmp_path), "out%d" % exact) + "/"
        p = subprocess.run([runner, ypath, out], cwd=h.REPO, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-2000:]
        assert "USE_SO3 ICP: 0" in p.stdout
        rows = h.read_csv_rows(out + "all_results.csv")
        assert [r["Method"] for r in rows] == ["FCN-SR", "ME-SR", "ME-TReg", "ME-TSVD"]
        ocfg = po.default_config(search_radius=1.0, max_iterations=30, kappa_target=10.0, std_reg_gamma=100.0, euler_exact_jacobian=exact)
        ores, ologs, _ = po.icp_run_euler(po.KdTree(pts), pts, (0.0, 0.0, 0.0, 0.01, 0.01, 0.01), "ME-SR", ocfg)
        me = [r for r in rows if r["Method"] == "ME-SR"][0]
        assert int(me["Iterations"]) == ores.iterations and int(me["Converged"]) == ores.converged == exact
        if exact:      # (the reference's row wanders for 30 iterations: only a contracting run is compared to the digit)
            assert abs(float(me["Trans_Error_m"]) - ologs[-1].trans_err) < 1e-6
