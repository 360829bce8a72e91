"""Property tests (hypothesis) of the host solver seam: algebraic identities that must hold for any SPD system."""
import numpy as np
from hypothesis import given, settings, strategies as st

from dcreg_amd import api


def _spd(seed, weak):
    rng = np.random.default_rng(seed)
    A = rng.normal(size=(80, 6)) * np.array([20, 20, 20, 1, 1, weak])
    b = rng.normal(size=80)
    return A.T @ A, A.T @ b


CFG = dict(search_radius=1.0, KAPPA_TARGET=10.0, STD_REG_GAMMA=100.0, DEGENERACY_THRES_COND=10.0, DEGENERACY_THRES_EIG=120.0,
           always_compute_schur=1)


@settings(max_examples=60, deadline=None)
@given(st.integers(0, 10_000), st.floats(0.01, 1.0))
def test_solver_identities(seed, weak):
    H, g = _spd(seed, weak)
    cfg = api.default_config(**CFG)
    # NONE: plain Gauss-Newton step
    an = api.analyze_degeneracy(H, "NONE_DETE", "NONE_HAND", cfg)
    x = api.solve_degenerate_system(H, g, "NONE_HAND", cfg, an)
    assert np.allclose(H @ x, g, rtol=1e-8, atol=1e-8 * np.abs(g).max())
    ev = np.array(an.eigenvalues_full[:])
    assert np.all(np.diff(ev) >= 0) and np.allclose(ev, np.linalg.eigvalsh(H), rtol=1e-9)
    V = np.array(an.eigenvectors_full[:]).reshape(6, 6)
    assert np.allclose(V.T @ V, np.eye(6), atol=1e-10) and np.allclose(H @ V, V * ev, rtol=1e-8, atol=1e-8 * ev[-1])
    assert np.allclose(sorted(an.singular_values[:], reverse=True), an.singular_values[:])
    # TReg: (H + gamma I) x = g when degenerate
    an = api.analyze_degeneracy(H, "FULL_EVD_MIN_EIGENVALUE", "STANDARD_REGULARIZATION", cfg)
    x = api.solve_degenerate_system(H, g, "STANDARD_REGULARIZATION", cfg, an)
    Hr = H + (100.0 * np.eye(6) if an.isDegenerate else 0)
    assert np.allclose(Hr @ x, g, rtol=1e-8, atol=1e-8 * np.abs(g).max())
    # SR: the step has no component along masked eigenvectors and equals the projected GN step
    an = api.analyze_degeneracy(H, "FULL_EVD_MIN_EIGENVALUE", "SOLUTION_REMAPPING", cfg)
    x = api.solve_degenerate_system(H, g, "SOLUTION_REMAPPING", cfg, an)
    mask = np.array(an.degenerate_mask[:], bool)
    gn = np.linalg.solve(H, g)
    if mask.all():
        assert np.all(x == 0)
    else:
        P = V[:, ~mask] @ V[:, ~mask].T
        assert np.allclose(x, P @ gn, rtol=1e-7, atol=1e-10 * max(1.0, np.abs(gn).max()))
        assert np.allclose(V[:, mask].T @ x, 0, atol=1e-9 * max(1.0, np.abs(gn).max()))
    # Schur + PCG: Schur complements / preconditioner structure, PCG converges to the GN step
    an = api.analyze_degeneracy(H, "SCHUR_CONDITION_NUMBER", "PRECONDITIONED_CG", cfg)
    SR = H[:3, :3] - H[:3, 3:] @ np.linalg.solve(H[3:, 3:], H[3:, :3])
    ST = H[3:, 3:] - H[3:, :3] @ np.linalg.solve(H[:3, :3], H[:3, 3:])
    assert np.allclose(an.lambda_schur_rot[:], np.linalg.eigvalsh(SR), rtol=1e-8)
    assert np.allclose(an.lambda_schur_trans[:], np.linalg.eigvalsh(ST), rtol=1e-8)
    lr, lt = np.array(an.lambda_schur_rot[:]), np.array(an.lambda_schur_trans[:])
    assert list(an.degenerate_mask[:]) == list((lr[-1] / lr > 10).astype(int)) + list((lt[-1] / lt > 10).astype(int))
    P = np.array(an.P_preconditioner[:]).reshape(6, 6)
    assert np.allclose(P, P.T, atol=1e-14) and np.all(P[:3, 3:] == 0)
    assert np.allclose(np.linalg.eigvalsh(P[3:, 3:]), sorted(1.0 / np.maximum(lt, lt[-1] / 10.0)), rtol=1e-8)
    x = api.solve_degenerate_system(H, g, "PRECONDITIONED_CG", cfg, an)
    if an.isDegenerate:
        assert np.linalg.norm(H @ x - g) <= 2e-6 * np.linalg.norm(g) or an.pcg_iterations == 10
    else:
        assert np.allclose(x, gn, rtol=1e-8, atol=1e-12)


@settings(max_examples=40, deadline=None)
@given(st.lists(st.floats(-1.0, 1.0), min_size=6, max_size=6), st.lists(st.floats(-3.0, 3.0), min_size=6, max_size=6))
def test_boxplus_is_right_multiplication(dx, p):
    T0 = api.pose6d_to_matrix(p[0], p[1], p[2], p[3], p[4], p[5])
    R1, t1 = api.boxplus(T0[:3, :3], T0[:3, 3], np.array(dx))
    assert np.allclose(R1.T @ R1, np.eye(3), atol=1e-12) and np.isclose(np.linalg.det(R1), 1.0, atol=1e-12)
    assert np.allclose(t1, T0[:3, 3] + T0[:3, :3] @ np.array(dx[3:]), atol=1e-14)
    th = np.linalg.norm(dx[:3])
    dR = T0[:3, :3].T @ R1
    ang = np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1))
    assert abs(ang - th) < 1e-7 or th < 1e-7
    te, re_ = api.pose_error(T0, np.block([[R1, t1[:, None]], [np.zeros((1, 3)), np.ones((1, 1))]]))
    assert abs(re_ - np.degrees(th)) < 1e-6 and abs(te - np.linalg.norm(dx[3:])) < 1e-12
