// SE(3) state helpers of the engine: MathUtils::exp / SE3State::boxplus (DCReg/include/math_utils.hpp:20-33,
// 158-166), Pose6D2Matrix / calculatePoseError (DCReg/include/utils.hpp:452-460, 497-535).
#pragma once
#include <cmath>
#include <cstring>

namespace dcreg {

inline void mat3mul(const double *A, const double *B, double *C) {
    double o[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)
        o[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
    std::memcpy(C, o, sizeof(o));
}

// so(3) -> SO(3), Rodrigues; first-order for |w| < 1e-10 exactly like the reference
inline void so3Exp(const double w[3], double R[9]) {
    const double th = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    if (th < 1e-10) {
        const double S[9] = {1, -w[2], w[1], w[2], 1, -w[0], -w[1], w[0], 1};
        std::memcpy(R, S, sizeof(S));
        return;
    }
    const double ax = w[0] / th, ay = w[1] / th, az = w[2] / th;
    const double K[9] = {0, -az, ay, az, 0, -ax, -ay, ax, 0};
    double KK[9];
    mat3mul(K, K, KK);
    const double s = std::sin(th), c1 = 1.0 - std::cos(th);
    for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0 ? 1.0 : 0.0) + s * K[i] + c1 * KK[i];
}

// right perturbation: R <- R exp(w), t <- t + R v
inline void boxplus(const double R[9], const double t[3], const double dx[6], double Ro[9], double to[3]) {
    double E[9], Rn[9], tn[3];
    so3Exp(dx, E);
    mat3mul(R, E, Rn);
    for (int i = 0; i < 3; ++i) tn[i] = t[i] + R[i * 3] * dx[3] + R[i * 3 + 1] * dx[4] + R[i * 3 + 2] * dx[5];
    std::memcpy(Ro, Rn, sizeof(Rn));
    std::memcpy(to, tn, sizeof(tn));
}

// Translation * Rz(yaw) * Ry(pitch) * Rx(roll), row-major 4x4
inline void pose6dToMatrix(double roll, double pitch, double yaw, double x, double y, double z, double T[16]) {
    const double cr = std::cos(roll), sr = std::sin(roll), cp = std::cos(pitch), sp = std::sin(pitch);
    const double cy = std::cos(yaw), sy = std::sin(yaw);
    const double Rx[9] = {1, 0, 0, 0, cr, -sr, 0, sr, cr};
    const double Ry[9] = {cp, 0, sp, 0, 1, 0, -sp, 0, cp};
    const double Rz[9] = {cy, -sy, 0, sy, cy, 0, 0, 0, 1};
    double A[9], B[9];
    mat3mul(Rz, Ry, A);
    mat3mul(A, Rx, B);
    for (int i = 0; i < 16; ++i) T[i] = 0.0;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T[i * 4 + j] = B[i * 3 + j];
    T[3] = x; T[7] = y; T[11] = z; T[15] = 1.0;
}

inline void stateToMatrix(const double R[9], const double t[3], double T[16]) {
    for (int i = 0; i < 16; ++i) T[i] = 0.0;
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) T[i * 4 + j] = R[i * 3 + j]; T[i * 4 + 3] = t[i]; }
    T[15] = 1.0;
}

// E = gt^-1 * T ; translation norm ; rotation angle as Eigen::AngleAxisd(R_E).angle() (through a quaternion)
inline void poseError(const double gt[16], const double T[16], double *trans, double *rotDeg) {
    double E[9], te[3];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s += gt[k * 4 + i] * T[k * 4 + j];
            E[i * 3 + j] = s;
        }
        double s = 0.0;
        for (int k = 0; k < 3; ++k) s += gt[k * 4 + i] * (T[k * 4 + 3] - gt[k * 4 + 3]);
        te[i] = s;
    }
    *trans = std::sqrt(te[0] * te[0] + te[1] * te[1] + te[2] * te[2]);
    double q[4];   // w x y z
    const double tr = E[0] + E[4] + E[8];
    if (tr > 0.0) {
        double s = std::sqrt(tr + 1.0);
        q[0] = 0.5 * s; s = 0.5 / s;
        q[1] = (E[7] - E[5]) * s; q[2] = (E[2] - E[6]) * s; q[3] = (E[3] - E[1]) * s;
    } else {
        int i = 0;
        if (E[4] > E[0]) i = 1;
        if (E[8] > E[i * 4]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        double s = std::sqrt(E[i * 4] - E[j * 4] - E[k * 4] + 1.0);
        q[1 + i] = 0.5 * s; s = 0.5 / s;
        q[0] = (E[k * 3 + j] - E[j * 3 + k]) * s;
        q[1 + j] = (E[j * 3 + i] + E[i * 3 + j]) * s;
        q[1 + k] = (E[k * 3 + i] + E[i * 3 + k]) * s;
    }
    const double n = std::sqrt(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const double ang = n != 0.0 ? 2.0 * std::atan2(n, std::fabs(q[0])) : 0.0;
    *rotDeg = std::fabs(ang) * 180.0 / M_PI;
}

}  // namespace dcreg
