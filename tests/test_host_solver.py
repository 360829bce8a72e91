"""CPU tests of the product's host side (solver seam + SE(3) helpers) through the C-ABI: the library must
load without a GPU, export every declared symbol, and agree with the oracle and the golden traces."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import helpers as h
from dcreg_amd import api
from oracle import pyoracle as po

METHODS = ["ME-SR", "ME-TSVD", "ME-TReg", "FCN-SR", "Ours", "NONE"]


def test_library_loads_and_exports_every_declared_symbol():
    L = api.load()
    hdr = open(os.path.join(h.REPO, "include", "dcreg.h")).read() + open(os.path.join(h.REPO, "include", "dcreg_debug.h")).read()
    declared = set(re.findall(r"\b(dcreg_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(api.EXPORTS), declared ^ set(api.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name
    assert b"gfx950" in L.dcreg_version()


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(api.DcregError):
        api.Context(0)


def _cfgs(paper):
    kw = dict(search_radius=1.0, max_iterations=30, KAPPA_TARGET=10.0, STD_REG_GAMMA=100.0,
              DEGENERACY_THRES_COND=10.0, DEGENERACY_THRES_EIG=120.0,
              use_weight_derivative=int(paper), always_compute_schur=int(paper))
    okw = dict(search_radius=1.0, max_iterations=30, kappa_target=10.0, std_reg_gamma=100.0, thres_cond=10.0,
               thres_eig=120.0, use_weight_derivative=int(paper), always_compute_schur=int(paper))
    return api.default_config(**kw), po.default_config(**okw)


@pytest.fixture(scope="module")
def systems():
    """(H, g) pairs from the oracle on the fixture at several poses + synthetic SPD systems."""
    pts = h.cylinder_cloud()
    tree = po.KdTree(pts)
    out = []
    for init, wd in ((h.RELEASE_INIT, 0), (h.PAPER_INIT, 1)):
        T0 = h.pose6d_matrix(**init)
        lo = po.linearize(tree, pts, T0[:3, :3], T0[:3, 3], po.default_lin_params(1.0, wd))
        out.append((lo["H"], lo["g"]))
    rng = np.random.default_rng(7)
    for k in range(40):
        A = rng.normal(size=(60, 6)) * np.array([30, 30, 30, 1, 1, 0.05 + 0.2 * (k % 5)])
        b = rng.normal(size=60)
        out.append((A.T @ A, A.T @ b))
    return out


@pytest.mark.parametrize("paper", [False, True])
@pytest.mark.parametrize("method", METHODS)
def test_analyze_and_solve_match_oracle(method, paper, systems):
    cfg, ocfg = _cfgs(paper)
    det, hand = api.METHODS[method]
    for H, g in systems:
        an = api.analyze_degeneracy(H, det, hand, cfg)
        oan = po.analyze(H, det, hand, ocfg)
        assert an.isDegenerate == oan.is_degenerate
        assert list(an.degenerate_mask[:]) == list(oan.mask[:])
        assert np.allclose(an.eigenvalues_full[:], oan.eigenvalues_full[:], rtol=1e-10, atol=1e-9)
        assert np.allclose(an.singular_values[:], oan.singular_values[:], rtol=1e-10, atol=1e-9)
        for f in ("cond_full", "cond_full_sub_rot", "cond_full_sub_trans", "cond_schur_rot", "cond_schur_trans",
                  "cond_diag_rot", "cond_diag_trans"):
            a, b = getattr(an, f), getattr(oan, f)
            assert (np.isnan(a) and np.isnan(b)) or np.isclose(a, b, rtol=1e-8), f
        if not np.isnan(oan.cond_schur_rot):
            assert np.allclose(an.lambda_schur_rot[:], oan.lambda_schur_rot[:], rtol=1e-9)
            assert np.allclose(an.lambda_schur_trans[:], oan.lambda_schur_trans[:], rtol=1e-9)
            assert np.allclose(np.array(an.P_preconditioner[:]), np.array(oan.P_preconditioner[:]), rtol=1e-8, atol=1e-14)
        x = api.solve_degenerate_system(H, g, hand, cfg, an)
        ox = po.solve(H, g, hand, ocfg, oan)
        assert np.allclose(x, ox, rtol=1e-8, atol=1e-12 * max(1.0, np.max(np.abs(ox)))), (method, x, ox)


@pytest.mark.parametrize("paper", [False, True])
def test_two_part_analysis_writes_the_same_record(paper, systems):
    """The pipelined engine takes the part of the analysis its step needs first and pays the rest - the 6x6 eigen-decomposition and the
    eigenvalues of the diagonal blocks, diagnostics of the log for Schur detection + PCG - after the device has its new pose.  Both
    parts together must write byte for byte what dcreg_analyze_degeneracy writes, for every method (those whose step reads the
    eigen-decomposition defer nothing)."""
    cfg, _ = _cfgs(paper)
    deferred = 0
    for method in METHODS:
        det, hand = api.METHODS[method]
        for H, _g in systems:
            one = api.analyze_degeneracy(H, det, hand, cfg)
            two, owed = api.analyze_degeneracy_two_part(H, det, hand, cfg)
            assert bytes(one) == bytes(two), (method, owed)
            deferred += owed != 0
    assert deferred > 0


def test_ours_first_iteration_golden():
    """degeneracy_analysis_first_iter.txt (paper run): spectra, kappas, mask, alignment, dx = Gauss-Newton."""
    pts = h.cylinder_cloud()
    tree = po.KdTree(pts)
    T0 = h.pose6d_matrix(**h.PAPER_INIT)
    lo = po.linearize(tree, pts, T0[:3, :3], T0[:3, 3], po.default_lin_params(1.0, 1))
    cfg, _ = _cfgs(True)
    an = api.analyze_degeneracy(lo["H"], "SCHUR_CONDITION_NUMBER", "PRECONDITIONED_CG", cfg)
    assert np.allclose(an.lambda_schur_rot[:], [422.505477, 1447.735216, 2999.323349], rtol=1e-6)
    assert np.allclose(an.lambda_schur_trans[:], [0.629416, 5.601848, 16.871859], rtol=1e-6)
    assert list(an.degenerate_mask[:]) == [0, 0, 0, 1, 0, 0]
    # Alignment Analysis block: aligned axis j <- Schur eigenvector orig_idx
    assert list(an.rot_indices[:]) == [0, 2, 1]
    assert list(an.trans_indices[:]) == [2, 1, 0]
    Vr = np.array(an.aligned_V_rot[:]).reshape(3, 3)
    Vt = np.array(an.aligned_V_trans[:]).reshape(3, 3)
    ang = lambda V, j: np.degrees(np.arccos(min(1.0, abs(V[j, j]))))
    # printed angles use the raw eigenvectors; Gram-Schmidt moves them by < 1e-9 deg for an orthonormal basis
    assert abs(ang(Vr, 0) - 11.821719) < 1e-3 and abs(ang(Vt, 2) - 0.434531) < 1e-3
    assert np.allclose(Vr.T @ Vr, np.eye(3), atol=1e-12) and np.allclose(Vt.T @ Vt, np.eye(3), atol=1e-12)
    x = api.solve_degenerate_system(lo["H"], lo["g"], "PRECONDITIONED_CG", cfg, an)
    gold_dx = [0.03422220, -0.00921189, -0.01426251, -0.12247351, -0.25354587, -1.05963507]
    assert np.max(np.abs(x - gold_dx)) < 1.5e-7
    assert 1 <= an.pcg_iterations <= 10


def test_tsvd_quirk_mask_indexes_descending_sigma():
    """dcreg.hpp:223-248: mask[0] (smallest eigenvalue) removes the LARGEST singular direction."""
    H = np.diag([1e4, 2e4, 3e4, 500.0, 300.0, 10.0])
    g = np.ones(6)
    cfg, _ = _cfgs(False)
    an = api.analyze_degeneracy(H, "FULL_EVD_MIN_EIGENVALUE", "TRUNCATED_SVD", cfg)
    assert list(an.degenerate_mask[:]) == [1, 0, 0, 0, 0, 0]
    x = api.solve_degenerate_system(H, g, "TRUNCATED_SVD", cfg, an)
    assert np.allclose(x, [1e-4, 0.5e-4, 0.0, 1 / 500.0, 1 / 300.0, 0.1])


def test_se3_helpers_match_oracle():
    rng = np.random.default_rng(3)
    L = po.lib()
    for _ in range(50):
        w = rng.normal(size=3) * rng.choice([1e-12, 1e-3, 0.5])
        dx = np.concatenate([w, rng.normal(size=3)])
        R0 = h.pose6d_matrix(0, 0, 0, *rng.normal(size=3))[:3, :3]
        t0 = rng.normal(size=3)
        R1, t1 = api.boxplus(R0, t0, dx)
        oR, ot = np.empty(9), np.empty(3)
        L.orc_boxplus(po._dp(np.ascontiguousarray(R0).reshape(9)), po._dp(t0), po._dp(dx), po._dp(oR), po._dp(ot))
        assert np.allclose(R1.reshape(9), oR, atol=1e-15) and np.allclose(t1, ot, atol=1e-15)
        T = np.eye(4); T[:3, :3] = R1; T[:3, 3] = t1
        gt = h.pose6d_matrix(*rng.normal(size=3), *(rng.normal(size=3) * 0.3))
        assert np.allclose(api.pose_error(gt, T), po.pose_error(gt, T), rtol=1e-12, atol=1e-14)
    p = rng.normal(size=6)
    assert np.allclose(api.pose6d_to_matrix(*p), po.pose6d_to_matrix(*p), atol=1e-15)
    assert np.allclose(api.pose6d_to_matrix(p[0], p[1], p[2], p[3], p[4], p[5]),
                       h.pose6d_matrix(p[3], p[4], p[5], p[0], p[1], p[2]), atol=1e-15)
