// Host side of the solver seam: DCReg::analyzeDegeneracy / solveDegenerateSystem
// (DCReg/include/dcreg.hpp:45-264), the Schur-complement block of the Euler engine
// (DCReg/src/icp_test_runner.cpp:2418-2469) and the pieces the released source leaves as stubs
// (SCHUR_CONDITION_NUMBER detection dcreg.hpp:96-98, PCG dcreg.hpp:186-193,279-287, axis alignment
// dcreg.hpp:267-276), written from the paper's description and pinned by the committed "Ours" traces
// (results/simulation/table3_fig9_fig10 of the reference).  Pure functions of (H, g, config); no device involved.
#include <cmath>
#include <cstring>
#include <limits>

#include "../../../include/dcreg.h"
#include "linalg.hpp"
#include "se3.hpp"

namespace dcreg {

static Mat6 toMat6(const double H[36]) { Mat6 m; std::memcpy(m.v, H, sizeof(m.v)); return m; }
static Mat3 block3(const Mat6 &H, int r0, int c0) {
    Mat3 b;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) b(i, j) = H(r0 + i, c0 + j);
    return b;
}
static double vmin3(const double *v) { return std::min(v[0], std::min(v[1], v[2])); }
static double vmax3(const double *v) { return std::max(v[0], std::max(v[1], v[2])); }

// Assign each Schur eigenvector to the physical axis it is closest to, fix signs, re-orthonormalise.
// aligned column j <-> reference axis e_j ; indices[j] = column of V_raw that was used.
static void alignAndOrthonormalize(const double Vraw[9], double Valigned[9], int indices[3]) {
    bool usedAxis[3] = {false, false, false}, usedVec[3] = {false, false, false};
    for (int round = 0; round < 3; ++round) {
        int bj = -1, bk = -1; double best = -1.0;
        for (int j = 0; j < 3; ++j) if (!usedAxis[j])
            for (int k = 0; k < 3; ++k) if (!usedVec[k]) {
                double a = std::fabs(Vraw[j * 3 + k]);          // |v_k . e_j|
                if (a > best) { best = a; bj = j; bk = k; }
            }
        usedAxis[bj] = usedVec[bk] = true;
        indices[bj] = bk;
    }
    double cols[3][3];
    for (int j = 0; j < 3; ++j) {
        const int k = indices[j];
        const double sgn = Vraw[j * 3 + k] < 0.0 ? -1.0 : 1.0;
        for (int r = 0; r < 3; ++r) cols[j][r] = sgn * Vraw[r * 3 + k];
    }
    for (int j = 0; j < 3; ++j) {   // Gram-Schmidt in axis order
        for (int p = 0; p < j; ++p) {
            double d = 0.0;
            for (int r = 0; r < 3; ++r) d += cols[j][r] * cols[p][r];
            for (int r = 0; r < 3; ++r) cols[j][r] -= d * cols[p][r];
        }
        double n = std::sqrt(cols[j][0] * cols[j][0] + cols[j][1] * cols[j][1] + cols[j][2] * cols[j][2]);
        if (n > 0.0) for (int r = 0; r < 3; ++r) cols[j][r] /= n;
    }
    for (int r = 0; r < 3; ++r) for (int j = 0; j < 3; ++j) Valigned[r * 3 + j] = cols[j][r];
}

// icp_test_runner.cpp:2418-2469 + eigenvalue-clamped block preconditioner
// eigenvalues and condition numbers of the diagonal blocks (the first lines of the Schur analysis): read by the EVD_SUB_CONDITION
// detection only, otherwise a diagnostic of the log
static void diagBlocks(const Mat6 &H, dcreg_analysis &res) {
    const Mat3 Hrr = block3(H, 0, 0), Htt = block3(H, 3, 3);
    Vec<3> w; Mat3 V;
    symEig<3>(Hrr, w, V);
    std::memcpy(res.lambda_sub_rot, w.data(), sizeof(double) * 3);
    res.cond_diag_rot = vmax3(res.lambda_sub_rot) / std::max(vmin3(res.lambda_sub_rot), 1e-12);
    symEig<3>(Htt, w, V);
    std::memcpy(res.lambda_sub_trans, w.data(), sizeof(double) * 3);
    res.cond_diag_trans = vmax3(res.lambda_sub_trans) / std::max(vmin3(res.lambda_sub_trans), 1e-12);
}
// defer_diag: the diagonal blocks and the axis alignment of the eigenvectors (read by the report writers only) are left to analyzeFinish;
// returns whether the alignment is owed (the analysis got that far)
static bool schurAnalysis(const Mat6 &H, const dcreg_config &cfg, dcreg_analysis &res, bool defer_diag = false) {
    const Mat3 Hrr = block3(H, 0, 0), Htt = block3(H, 3, 3), Hrt = block3(H, 0, 3), Htr = block3(H, 3, 0);
    if (!defer_diag) diagBlocks(H, res);

    Mat3 HttInv, HrrInv;
    const bool okT = fullPivLuInverse3(Htt, HttInv), okR = fullPivLuInverse3(Hrr, HrrInv);
    if (!(okT && okR)) {   // :2464-2469
        res.cond_schur_rot = res.cond_schur_trans = std::numeric_limits<double>::infinity();
        return false;
    }
    Mat3 SR = mul(mul(Hrt, HttInv), Htr), ST = mul(mul(Htr, HrrInv), Hrt);
    for (int i = 0; i < 9; ++i) { SR.v[i] = Hrr.v[i] - SR.v[i]; ST.v[i] = Htt.v[i] - ST.v[i]; }
    Vec<3> lr, lt; Mat3 Vr, Vt;
    const bool e1 = symEig<3>(SR, lr, Vr), e2 = symEig<3>(ST, lt, Vt);
    if (!(e1 && e2)) {     // :2460-2463
        res.cond_schur_rot = res.cond_schur_trans = std::numeric_limits<double>::infinity();
        return false;
    }
    std::memcpy(res.lambda_schur_rot, lr.data(), sizeof(double) * 3);
    std::memcpy(res.lambda_schur_trans, lt.data(), sizeof(double) * 3);
    std::memcpy(res.schur_V_rot, Vr.v, sizeof(Vr.v));
    std::memcpy(res.schur_V_trans, Vt.v, sizeof(Vt.v));
    res.cond_schur_rot = vmax3(res.lambda_schur_rot) / std::max(vmin3(res.lambda_schur_rot), 1e-12);
    res.cond_schur_trans = vmax3(res.lambda_schur_trans) / std::max(vmin3(res.lambda_schur_trans), 1e-12);
    if (!defer_diag) {
        alignAndOrthonormalize(res.schur_V_rot, res.aligned_V_rot, res.rot_indices);
        alignAndOrthonormalize(res.schur_V_trans, res.aligned_V_trans, res.trans_indices);
    }
    // P = blockdiag(V_R diag(1/max(l, lmax/kappa_tg)) V_R^T, same for t)
    for (int blk = 0; blk < 2; ++blk) {
        const double *lam = blk ? res.lambda_schur_trans : res.lambda_schur_rot;
        const double *Vb = blk ? res.schur_V_trans : res.schur_V_rot;
        const double floorL = vmax3(lam) / cfg.KAPPA_TARGET;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s += Vb[i * 3 + k] * Vb[j * 3 + k] / std::max(lam[k], floorL);
            res.P_preconditioner[(3 * blk + i) * 6 + 3 * blk + j] = s;
        }
    }
    return defer_diag;
}

// full eigen-decomposition block of the analysis (dcreg.hpp:66-89): eigenvalues / eigenvectors / singular values / condition
// numbers of H.  Only the FULL_EVD / SUB_CONDITION / FULL_SVD detections and the remapping / truncated-SVD handlings read them
// back; for the others (Schur detection + PCG: "Ours") they are diagnostics of the iteration log.
static bool fullEvdBlock(const Mat6 &H, dcreg_analysis &res) {
    const double nan = std::numeric_limits<double>::quiet_NaN();
    Vec<6> ev; Mat6 V;
    const bool evdOk = symEig<6>(H, ev, V);
    if (evdOk) {
        std::memcpy(res.eigenvalues_full, ev.data(), sizeof(double) * 6);
        std::memcpy(res.eigenvectors_full, V.v, sizeof(V.v));
        res.cond_full_sub_trans = std::fabs(ev[2]) / std::max(std::fabs(ev[0]), 1e-12);
        res.cond_full_sub_rot = std::fabs(ev[5]) / std::max(std::fabs(ev[3]), 1e-12);
    } else {
        res.cond_full_sub_rot = res.cond_full_sub_trans = std::numeric_limits<double>::infinity();
        for (double &x : res.eigenvalues_full) x = nan;
    }
    // dcreg.hpp:83-89: JacobiSVD of a symmetric matrix: sigma = |lambda|, descending
    for (int i = 0; i < 6; ++i) res.singular_values[i] = std::fabs(res.eigenvalues_full[i]);
    std::sort(res.singular_values, res.singular_values + 6, [](double a, double b) { return a > b; });
    res.cond_full = res.singular_values[5] > 1e-12 ? res.singular_values[0] / res.singular_values[5]
                                                     : std::numeric_limits<double>::infinity();
    return evdOk;
}
// can the step (mask, solve) of this detection / handling pair be taken before the full eigen-decomposition exists?
static bool evdIsDiagnosticOnly(int detection, int handling) {
    const bool det = detection == DCREG_NONE_DETE || detection == DCREG_SCHUR_CONDITION_NUMBER;
    const bool hand = handling != DCREG_SOLUTION_REMAPPING && handling != DCREG_TRUNCATED_SVD;
    return det && hand;
}

// defer_evd: leave the full eigen-decomposition block - and the diagonal blocks of the Schur analysis - out when nothing of the step
// depends on them; analyzeFinish() then completes the record, bit for bit what the one-pass analysis writes.
// (returns a mask of what is owed: 1 = the full eigen-decomposition block, 2 = the diagonal blocks of the Schur analysis, 4 = the axis
//  alignment of the Schur eigenvectors)
static int analyze(const Mat6 &H, int detection, int handling, const dcreg_config &cfg, dcreg_analysis &res, bool defer_evd = false) {
    const double nan = std::numeric_limits<double>::quiet_NaN();
    std::memset(&res, 0, sizeof(res));
    res.cond_schur_rot = res.cond_schur_trans = res.cond_diag_rot = res.cond_diag_trans = nan;
    for (int i = 0; i < 3; ++i) {
        res.lambda_schur_rot[i] = res.lambda_schur_trans[i] = res.lambda_sub_rot[i] = res.lambda_sub_trans[i] = nan;
        res.rot_indices[i] = res.trans_indices[i] = i;
    }
    for (int i = 0; i < 6; ++i) res.P_preconditioner[i * 7] = 1.0;
    for (int i = 0; i < 3; ++i) res.schur_V_rot[i * 4] = res.schur_V_trans[i * 4] = res.aligned_V_rot[i * 4] = res.aligned_V_trans[i * 4] = 1.0;

    const bool deferred = defer_evd && evdIsDiagnosticOnly(detection, handling);
    const bool evdOk = deferred ? false : fullEvdBlock(H, res);      // dcreg.hpp:66-89

    const bool schur = detection == DCREG_SCHUR_CONDITION_NUMBER || handling == DCREG_PRECONDITIONED_CG || cfg.always_compute_schur;
    const bool align_owed = schur && schurAnalysis(H, cfg, res, deferred);

    switch (detection) {
    case DCREG_SCHUR_CONDITION_NUMBER: {
        // per-direction rule consistent with every committed "Ours" iteration:
        // mask[i] = lmax(S_R)/l_i(S_R) > kappa_th (i<3), mask[3+i] = lmax(S_t)/l_i(S_t) > kappa_th
        if (std::isfinite(res.cond_schur_rot) && std::isfinite(res.cond_schur_trans)) {
            const double mr = vmax3(res.lambda_schur_rot), mt = vmax3(res.lambda_schur_trans);
            for (int i = 0; i < 3; ++i) {
                if (mr / std::max(res.lambda_schur_rot[i], 1e-12) > cfg.DEGENERACY_THRES_COND) res.degenerate_mask[i] = 1;
                if (mt / std::max(res.lambda_schur_trans[i], 1e-12) > cfg.DEGENERACY_THRES_COND) res.degenerate_mask[3 + i] = 1;
            }
            for (int m : res.degenerate_mask) res.isDegenerate |= m;
        } else {
            res.isDegenerate = 1;
            for (int &m : res.degenerate_mask) m = 1;
        }
        break;
    }
    case DCREG_FULL_EVD_MIN_EIGENVALUE:   // dcreg.hpp:100-110
        if (evdOk)
            for (int i = 0; i < 6; ++i)
                if (res.eigenvalues_full[i] < cfg.DEGENERACY_THRES_EIG) { res.isDegenerate = 1; res.degenerate_mask[i] = 1; }
        break;
    case DCREG_EVD_SUB_CONDITION:         // dcreg.hpp:112-126 (NaN unless the Schur block ran -> never fires in the release)
        res.isDegenerate = (res.cond_diag_rot > cfg.DEGENERACY_THRES_COND || res.cond_diag_trans > cfg.DEGENERACY_THRES_COND);
        if (res.isDegenerate) {
            if (res.cond_diag_trans > cfg.DEGENERACY_THRES_COND) for (int i = 0; i < 3; ++i) res.degenerate_mask[i + 3] = 1;
            if (res.cond_diag_rot > cfg.DEGENERACY_THRES_COND) for (int i = 0; i < 3; ++i) res.degenerate_mask[i] = 1;
        }
        break;
    case DCREG_FULL_SVD_CONDITION:        // dcreg.hpp:128-153
        res.isDegenerate = res.cond_full > cfg.DEGENERACY_THRES_COND;
        if (res.isDegenerate) {
            double mx = res.eigenvalues_full[0];
            for (double e : res.eigenvalues_full) mx = std::max(mx, e);
            for (int i = 0; i < 6; ++i)
                if (mx / res.eigenvalues_full[i] > cfg.DEGENERACY_THRES_COND) res.degenerate_mask[i] = 1;
        }
        break;
    default: break;                        // NONE_DETE and everything else: not degenerate
    }
    return (deferred ? 1 : 0) | ((deferred && schur) ? 2 : 0) | (align_owed ? 4 : 0);
}

// preconditioned conjugate gradients on the 6x6 SPD system (dcreg.hpp:279-287 is a stub)
static Vec<6> solvePCG(const Mat6 &A, const Vec<6> &b, const Mat6 &P, int maxIter, double tol, int &iters) {
    Vec<6> x{}; x.fill(0.0);
    Vec<6> r = b;
    const double bn = norm<6>(b);
    iters = 0;
    if (bn == 0.0) return x;
    Vec<6> z = mulv<6>(P, r), p = z;
    double rz = dot<6>(r, z);
    for (int k = 0; k < maxIter; ++k) {
        const Vec<6> Ap = mulv<6>(A, p);
        const double pAp = dot<6>(p, Ap);
        if (!(pAp > 0.0)) break;
        const double alpha = rz / pAp;
        for (int i = 0; i < 6; ++i) { x[i] += alpha * p[i]; r[i] -= alpha * Ap[i]; }
        iters = k + 1;
        if (norm<6>(r) <= tol * bn) break;
        z = mulv<6>(P, r);
        const double rzNew = dot<6>(r, z);
        const double beta = rzNew / rz;
        rz = rzNew;
        for (int i = 0; i < 6; ++i) p[i] = z[i] + beta * p[i];
    }
    return x;
}

static Vec<6> solve(const Mat6 &H, const Vec<6> &g, int handling, const dcreg_config &cfg, dcreg_analysis &an) {
    Vec<6> x;
    switch (handling) {
    case DCREG_STANDARD_REGULARIZATION: {   // dcreg.hpp:177-184
        Mat6 Hr = H;
        if (an.isDegenerate) for (int i = 0; i < 6; ++i) Hr(i, i) += cfg.STD_REG_GAMMA;
        colPivHouseholderQrSolve<6, 6>(Hr, g, x);
        return x;
    }
    case DCREG_PRECONDITIONED_CG:           // dcreg.hpp:186-193
        if (an.isDegenerate) {
            Mat6 P; std::memcpy(P.v, an.P_preconditioner, sizeof(P.v));
            return solvePCG(H, g, P, cfg.PCG_MAX_ITER, cfg.PCG_TOLERANCE, an.pcg_iterations);
        }
        colPivHouseholderQrSolve<6, 6>(H, g, x);
        return x;
    case DCREG_SOLUTION_REMAPPING: {        // dcreg.hpp:195-221
        colPivHouseholderQrSolve<6, 6>(H, g, x);
        bool finite = true;
        for (double e : an.eigenvalues_full) finite &= std::isfinite(e);
        if (an.isDegenerate && finite) {
            Vec<6> y{}; y.fill(0.0);
            int good = 0;
            for (int i = 0; i < 6; ++i) {
                if (an.degenerate_mask[i]) continue;
                ++good;
                double d = 0.0;
                for (int k = 0; k < 6; ++k) d += an.eigenvectors_full[k * 6 + i] * x[k];
                for (int k = 0; k < 6; ++k) y[k] += an.eigenvectors_full[k * 6 + i] * d;
            }
            if (good > 0) x = y; else x.fill(0.0);
        }
        return x;
    }
    case DCREG_TRUNCATED_SVD: {             // dcreg.hpp:223-248
        // sigma is descending while degenerate_mask is indexed by ascending eigenvalue: mask[0] (smallest
        // lambda) therefore removes the LARGEST singular direction.  The goldens reproduce this quirk.
        int order[6] = {0, 1, 2, 3, 4, 5};
        std::sort(order, order + 6, [&](int a, int b) {
            return std::fabs(an.eigenvalues_full[a]) > std::fabs(an.eigenvalues_full[b]); });
        Vec<6> y{}; y.fill(0.0);
        int retained = 0;
        for (int i = 0; i < 6; ++i) {
            const double sv = an.singular_values[i];
            if (an.degenerate_mask[i] || !(sv > 1e-9)) continue;
            ++retained;
            const int e = order[i];
            const double sgn = an.eigenvalues_full[e] < 0.0 ? -1.0 : 1.0;   // u_i = sign(lambda) v_i
            double d = 0.0;
            for (int k = 0; k < 6; ++k) d += sgn * an.eigenvectors_full[k * 6 + e] * g[k];
            for (int k = 0; k < 6; ++k) y[k] += an.eigenvectors_full[k * 6 + e] * d / sv;
        }
        if (retained == 0) y.fill(0.0);
        return y;
    }
    default:                                // NONE_HAND, ADAPTIVE_REGULARIZATION (no case) : dcreg.hpp:250-257
        colPivHouseholderQrSolve<6, 6>(H, g, x);
        return x;
    }
}

void analyzeDegeneracy(const double H[36], int detection, int handling, const dcreg_config &cfg, dcreg_analysis &res) {
    analyze(toMat6(H), detection, handling, cfg, res);
}
// the analysis in two parts, for a loop that wants the step out of the door first (engine.cpp): analyzeStep returns what is still
// owed (the eigen-decomposition block, the diagonal blocks of the Schur analysis), analyzeFinish pays it.  Together they write exactly what analyzeDegeneracy writes.
int analyzeStep(const double H[36], int detection, int handling, const dcreg_config &cfg, dcreg_analysis &res) {
    return analyze(toMat6(H), detection, handling, cfg, res, true);
}
void analyzeFinish(const double H[36], dcreg_analysis &res, int owed) {
    const Mat6 M = toMat6(H);
    if (owed & 1) fullEvdBlock(M, res);
    if (owed & 2) diagBlocks(M, res);
    if (owed & 4) {
        alignAndOrthonormalize(res.schur_V_rot, res.aligned_V_rot, res.rot_indices);
        alignAndOrthonormalize(res.schur_V_trans, res.aligned_V_trans, res.trans_indices);
    }
}
void solveDegenerateSystem(const double H[36], const double g[6], int handling, const dcreg_config &cfg,
                           dcreg_analysis &an, double x[6]) {
    Vec<6> gv; std::memcpy(gv.data(), g, sizeof(double) * 6);
    const Vec<6> xv = solve(toMat6(H), gv, handling, cfg, an);
    std::memcpy(x, xv.data(), sizeof(double) * 6);
}
bool invertSpd6(const double H[36], double inv[36]) {   // covariance: FullPivLU::isInvertible() + inverse(), icp_test_runner.cpp:2016-2018
    const Mat6 A = toMat6(H);
    Mat6 I;
    if (!fullPivLuInverse<6>(A, I)) return false;
    std::memcpy(inv, I.v, sizeof(I.v));
    return true;
}

}  // namespace dcreg

extern "C" {

void dcreg_default_config(dcreg_config *c) {
    std::memset(c, 0, sizeof(*c));
    c->search_radius = 1.0; c->max_iterations = 30;                        // utils.hpp:150-151
    c->CONVERGENCE_THRESH_ROT = 1e-5; c->CONVERGENCE_THRESH_TRANS = 1e-3;  // utils.hpp:139-140
    c->DEGENERACY_THRES_COND = 10.0; c->DEGENERACY_THRES_EIG = 120.0;      // utils.hpp:83-84
    c->KAPPA_TARGET = 1.0; c->PCG_TOLERANCE = 1e-6; c->PCG_MAX_ITER = 10;  // utils.hpp:85-87
    c->STD_REG_GAMMA = 0.01; c->ADAPTIVE_REG_ALPHA = 10.0;                 // utils.hpp:88-89
    for (int i = 0; i < 4; ++i) c->gt_matrix[i * 5] = 1.0;
}

// (dcreg_debug.h) the two-part analysis of the pipelined engine - step part, then what it left owed - for the parity tests
int dcreg_analyze_degeneracy_two_part(const double H[36], int detection, int handling, const dcreg_config *cfg, dcreg_analysis *res,
                                      int *owed) {
    if (!H || !cfg || !res) return DCREG_E_INVALID;
    const int o = dcreg::analyzeStep(H, detection, handling, *cfg, *res);
    if (owed) *owed = o;
    dcreg::analyzeFinish(H, *res, o);
    return DCREG_OK;
}
int dcreg_analyze_degeneracy(const double H[36], int detection, int handling, const dcreg_config *cfg, dcreg_analysis *res) {
    if (!H || !cfg || !res) return DCREG_E_INVALID;
    dcreg::analyzeDegeneracy(H, detection, handling, *cfg, *res);
    return DCREG_OK;
}

int dcreg_solve_degenerate_system(const double H[36], const double g[6], int handling, const dcreg_config *cfg,
                                  dcreg_analysis *an, double x[6]) {
    if (!H || !g || !cfg || !an || !x) return DCREG_E_INVALID;
    dcreg::solveDegenerateSystem(H, g, handling, *cfg, *an, x);
    return DCREG_OK;
}

void dcreg_unpack_hessian(const double U[21], double H[36]) {
    int idx = 0;
    for (int i = 0; i < 6; ++i) for (int j = i; j < 6; ++j) { H[i * 6 + j] = H[j * 6 + i] = U[idx++]; }
}

void dcreg_boxplus(const double R[9], const double t[3], const double dx[6], double Ro[9], double to[3]) {
    dcreg::boxplus(R, t, dx, Ro, to);
}
void dcreg_pose6d_to_matrix(double roll, double pitch, double yaw, double x, double y, double z, double T[16]) {
    dcreg::pose6dToMatrix(roll, pitch, yaw, x, y, z, T);
}
void dcreg_pose_error(const double gt[16], const double T[16], double *trans, double *rot) {
    dcreg::poseError(gt, T, trans, rot);
}

}  // extern "C"
