// Fixed-size dense algebra for the 6x6 / 3x3 host side (no Eigen on the box).
// Semantics follow the Eigen 3.3.7 routines the reference calls:
//   ColPivHouseholderQR (dcreg.hpp:182,190,197,245,251,255), SelfAdjointEigenSolver (dcreg.hpp:62,66;
//   icp_test_runner.cpp:2426-2449), JacobiSVD of a symmetric matrix (dcreg.hpp:63,83),
//   FullPivLU isInvertible/inverse (icp_test_runner.cpp:2422-2445).
#pragma once
#include <algorithm>
#include <array>
#include <cfloat>
#include <cmath>

namespace dcreg {

template <int R, int C>
struct Mat {
    double v[R * C];
    double &operator()(int r, int c) { return v[r * C + c]; }
    double operator()(int r, int c) const { return v[r * C + c]; }
    static Mat zero() { Mat m; for (double &x : m.v) x = 0.0; return m; }
    static Mat identity() { Mat m = zero(); for (int i = 0; i < (R < C ? R : C); ++i) m(i, i) = 1.0; return m; }
};
template <int N> using Vec = std::array<double, N>;
using Mat3 = Mat<3, 3>;
using Mat6 = Mat<6, 6>;

template <int R, int K, int C>
inline Mat<R, C> mul(const Mat<R, K> &a, const Mat<K, C> &b) {
    Mat<R, C> o;
    for (int i = 0; i < R; ++i) for (int j = 0; j < C; ++j) {
        double s = 0.0;
        for (int k = 0; k < K; ++k) s += a(i, k) * b(k, j);
        o(i, j) = s;
    }
    return o;
}
template <int R, int C>
inline Mat<C, R> transpose(const Mat<R, C> &a) {
    Mat<C, R> o;
    for (int i = 0; i < R; ++i) for (int j = 0; j < C; ++j) o(j, i) = a(i, j);
    return o;
}
template <int N>
inline Vec<N> mulv(const Mat<N, N> &a, const Vec<N> &x) {
    Vec<N> o;
    for (int i = 0; i < N; ++i) { double s = 0.0; for (int j = 0; j < N; ++j) s += a(i, j) * x[j]; o[i] = s; }
    return o;
}
template <int N> inline double dot(const Vec<N> &a, const Vec<N> &b) { double s = 0.0; for (int i = 0; i < N; ++i) s += a[i] * b[i]; return s; }
template <int N> inline double norm(const Vec<N> &a) { return std::sqrt(dot<N>(a, a)); }

// ---- symmetric eigen-decomposition: Householder tridiagonalisation + implicit-shift QL
//      (the scheme SelfAdjointEigenSolver uses).  Ascending eigenvalues, V columns = eigenvectors.
// sqrt(a^2 + b^2) of the QL sweeps.  Not std::hypot: its guard against overflow costs 30 ns a call, a dozen calls per 3x3 problem -
// two thirds of what a Monte-Carlo trial's host step took (the entries of H and of its Schur blocks are nowhere near 1e150)
inline double hyp2(double a, double b) { return std::sqrt(a * a + b * b); }
template <int N>
inline bool symEig(const Mat<N, N> &A, Vec<N> &w, Mat<N, N> &V) {
    double d[N], e[N];
    for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) V(i, j) = 0.5 * (A(i, j) + A(j, i));
    // tridiagonalise (tred2)
    for (int j = 0; j < N; ++j) d[j] = V(N - 1, j);
    for (int i = N - 1; i > 0; --i) {
        double scale = 0.0, h = 0.0;
        for (int k = 0; k < i; ++k) scale += std::fabs(d[k]);
        if (scale == 0.0) {
            e[i] = d[i - 1];
            for (int j = 0; j < i; ++j) { d[j] = V(i - 1, j); V(i, j) = 0.0; V(j, i) = 0.0; }
        } else {
            for (int k = 0; k < i; ++k) { d[k] /= scale; h += d[k] * d[k]; }
            double f = d[i - 1], g = std::sqrt(h);
            if (f > 0) g = -g;
            e[i] = scale * g;
            h -= f * g;
            d[i - 1] = f - g;
            for (int j = 0; j < i; ++j) e[j] = 0.0;
            for (int j = 0; j < i; ++j) {
                f = d[j];
                V(j, i) = f;
                g = e[j] + V(j, j) * f;
                for (int k = j + 1; k <= i - 1; ++k) { g += V(k, j) * d[k]; e[k] += V(k, j) * f; }
                e[j] = g;
            }
            f = 0.0;
            for (int j = 0; j < i; ++j) { e[j] /= h; f += e[j] * d[j]; }
            double hh = f / (h + h);
            for (int j = 0; j < i; ++j) e[j] -= hh * d[j];
            for (int j = 0; j < i; ++j) {
                f = d[j]; g = e[j];
                for (int k = j; k <= i - 1; ++k) V(k, j) -= (f * e[k] + g * d[k]);
                d[j] = V(i - 1, j);
                V(i, j) = 0.0;
            }
        }
        d[i] = h;
    }
    for (int i = 0; i < N - 1; ++i) {
        V(N - 1, i) = V(i, i);
        V(i, i) = 1.0;
        double h = d[i + 1];
        if (h != 0.0) {
            for (int k = 0; k <= i; ++k) d[k] = V(k, i + 1) / h;
            for (int j = 0; j <= i; ++j) {
                double g = 0.0;
                for (int k = 0; k <= i; ++k) g += V(k, i + 1) * V(k, j);
                for (int k = 0; k <= i; ++k) V(k, j) -= g * d[k];
            }
        }
        for (int k = 0; k <= i; ++k) V(k, i + 1) = 0.0;
    }
    for (int j = 0; j < N; ++j) { d[j] = V(N - 1, j); V(N - 1, j) = 0.0; }
    V(N - 1, N - 1) = 1.0;
    e[0] = 0.0;
    // QL with implicit shifts (tql2)
    for (int i = 1; i < N; ++i) e[i - 1] = e[i];
    e[N - 1] = 0.0;
    double f = 0.0, tst1 = 0.0;
    const double eps = DBL_EPSILON;
    bool ok = true;
    for (int l = 0; l < N; ++l) {
        tst1 = std::max(tst1, std::fabs(d[l]) + std::fabs(e[l]));
        int m = l;
        while (m < N) { if (std::fabs(e[m]) <= eps * tst1) break; m++; }
        if (m == N) m = N - 1;
        if (m > l) {
            int iter = 0;
            do {
                if (++iter > 60) { ok = false; break; }
                double g = d[l];
                double p = (d[l + 1] - g) / (2.0 * e[l]);
                double r = hyp2(p, 1.0);
                if (p < 0) r = -r;
                d[l] = e[l] / (p + r);
                d[l + 1] = e[l] * (p + r);
                double dl1 = d[l + 1];
                double h = g - d[l];
                for (int i = l + 2; i < N; ++i) d[i] -= h;
                f += h;
                p = d[m];
                double c = 1.0, c2 = c, c3 = c, el1 = e[l + 1], s = 0.0, s2 = 0.0;
                for (int i = m - 1; i >= l; --i) {
                    c3 = c2; c2 = c; s2 = s;
                    g = c * e[i];
                    h = c * p;
                    r = hyp2(p, e[i]);
                    e[i + 1] = s * r;
                    s = e[i] / r;
                    c = p / r;
                    p = c * d[i] - s * g;
                    d[i + 1] = h + s * (c * g + s * d[i]);
                    for (int k = 0; k < N; ++k) {
                        h = V(k, i + 1);
                        V(k, i + 1) = s * V(k, i) + c * h;
                        V(k, i) = c * V(k, i) - s * h;
                    }
                }
                p = -s * s2 * c3 * el1 * e[l] / dl1;
                e[l] = s * p;
                d[l] = c * p;
            } while (std::fabs(e[l]) > eps * tst1);
        }
        d[l] = d[l] + f;
        e[l] = 0.0;
    }
    // sort ascending
    for (int i = 0; i < N - 1; ++i) {
        int k = i; double p = d[i];
        for (int j = i + 1; j < N; ++j) if (d[j] < p) { k = j; p = d[j]; }
        if (k != i) {
            d[k] = d[i]; d[i] = p;
            for (int j = 0; j < N; ++j) std::swap(V(j, i), V(j, k));
        }
    }
    for (int i = 0; i < N; ++i) w[i] = d[i];
    return ok;
}

// ---- Eigen::ColPivHouseholderQR::solve for an M x N system (M >= N), returns nonzeroPivots()
template <int M, int N>
inline int colPivHouseholderQrSolve(const Mat<M, N> &A, const Vec<M> &b, Vec<N> &x) {
    Mat<M, N> qr = A;
    double hcoef[N], nrmUpd[N], nrmDir[N];
    int perm[N];
    double maxNorm = 0.0;
    for (int j = 0; j < N; ++j) {
        double s = 0.0;
        for (int i = 0; i < M; ++i) s += qr(i, j) * qr(i, j);
        nrmUpd[j] = nrmDir[j] = std::sqrt(s);
        maxNorm = std::max(maxNorm, nrmUpd[j]);
        perm[j] = j;
    }
    const double thresholdHelper = (maxNorm * DBL_EPSILON) * (maxNorm * DBL_EPSILON) / double(M);
    const double downdateThreshold = std::sqrt(DBL_EPSILON);
    int nonzeroPivots = N;
    for (int k = 0; k < N; ++k) {
        int best = k;
        for (int j = k + 1; j < N; ++j) if (nrmUpd[j] > nrmUpd[best]) best = j;
        if (nonzeroPivots == N && nrmUpd[best] * nrmUpd[best] < thresholdHelper * double(M - k)) nonzeroPivots = k;
        if (best != k) {
            for (int i = 0; i < M; ++i) std::swap(qr(i, k), qr(i, best));
            std::swap(nrmUpd[k], nrmUpd[best]);
            std::swap(nrmDir[k], nrmDir[best]);
            std::swap(perm[k], perm[best]);
        }
        // Householder vector for column k (Eigen makeHouseholder)
        double tailSq = 0.0;
        for (int i = k + 1; i < M; ++i) tailSq += qr(i, k) * qr(i, k);
        const double c0 = qr(k, k);
        double beta, tau;
        if (tailSq <= DBL_MIN) {
            tau = 0.0; beta = c0;
            for (int i = k + 1; i < M; ++i) qr(i, k) = 0.0;
        } else {
            beta = std::sqrt(c0 * c0 + tailSq);
            if (c0 >= 0.0) beta = -beta;
            const double denom = c0 - beta;
            for (int i = k + 1; i < M; ++i) qr(i, k) /= denom;
            tau = (beta - c0) / beta;
        }
        hcoef[k] = tau;
        qr(k, k) = beta;
        if (tau != 0.0)
            for (int j = k + 1; j < N; ++j) {
                double tmp = qr(k, j);
                for (int i = k + 1; i < M; ++i) tmp += qr(i, k) * qr(i, j);
                qr(k, j) -= tau * tmp;
                for (int i = k + 1; i < M; ++i) qr(i, j) -= tau * qr(i, k) * tmp;
            }
        for (int j = k + 1; j < N; ++j) {
            if (nrmUpd[j] == 0.0) continue;
            double t = std::fabs(qr(k, j)) / nrmUpd[j];
            t = (1.0 + t) * (1.0 - t);
            if (t < 0.0) t = 0.0;
            const double ratio = nrmUpd[j] / nrmDir[j];
            if (t * ratio * ratio <= downdateThreshold) {
                double s = 0.0;
                for (int i = k + 1; i < M; ++i) s += qr(i, j) * qr(i, j);
                nrmDir[j] = nrmUpd[j] = std::sqrt(s);
            } else {
                nrmUpd[j] *= std::sqrt(t);
            }
        }
    }
    for (int j = 0; j < N; ++j) x[j] = 0.0;
    if (nonzeroPivots == 0) return 0;
    Vec<M> c = b;
    for (int k = 0; k < nonzeroPivots; ++k) {
        if (hcoef[k] == 0.0) continue;
        double tmp = c[k];
        for (int i = k + 1; i < M; ++i) tmp += qr(i, k) * c[i];
        c[k] -= hcoef[k] * tmp;
        for (int i = k + 1; i < M; ++i) c[i] -= hcoef[k] * qr(i, k) * tmp;
    }
    for (int i = nonzeroPivots - 1; i >= 0; --i) {
        double s = c[i];
        for (int j = i + 1; j < nonzeroPivots; ++j) s -= qr(i, j) * c[j];
        c[i] = s / qr(i, i);
    }
    for (int i = 0; i < nonzeroPivots; ++i) x[perm[i]] = c[i];
    return nonzeroPivots;
}

// ---- Eigen::FullPivLU<MatrixNd>: isInvertible() + inverse()  (3x3 Schur blocks :2422-2445, 6x6 covariance :2016-2018)
template <int N>
inline bool fullPivLuInverse(const Mat<N, N> &A, Mat<N, N> &inv) {
    Mat<N, N> lu = A;
    int rowOf[N], colOf[N];
    double pivots[N], maxPivot = 0.0;
    for (int i = 0; i < N; ++i) { rowOf[i] = colOf[i] = i; pivots[i] = 0.0; }
    for (int k = 0; k < N; ++k) {
        int pr = k, pc = k; double best = -1.0;
        for (int i = k; i < N; ++i) for (int j = k; j < N; ++j)
            if (std::fabs(lu(i, j)) > best) { best = std::fabs(lu(i, j)); pr = i; pc = j; }
        if (best == 0.0) return false;
        maxPivot = std::max(maxPivot, best);
        if (pr != k) { for (int j = 0; j < N; ++j) std::swap(lu(k, j), lu(pr, j)); std::swap(rowOf[k], rowOf[pr]); }
        if (pc != k) { for (int i = 0; i < N; ++i) std::swap(lu(i, k), lu(i, pc)); std::swap(colOf[k], colOf[pc]); }
        pivots[k] = lu(k, k);
        for (int i = k + 1; i < N; ++i) {
            lu(i, k) /= lu(k, k);
            for (int j = k + 1; j < N; ++j) lu(i, j) -= lu(i, k) * lu(k, j);
        }
    }
    const double thr = DBL_EPSILON * double(N) * maxPivot;   // FullPivLU::threshold() * |maxpivot|
    for (double p : pivots) if (!(std::fabs(p) > thr)) return false;
    for (int col = 0; col < N; ++col) {
        double y[N], z[N];
        for (int i = 0; i < N; ++i) y[i] = rowOf[i] == col ? 1.0 : 0.0;
        for (int i = 0; i < N; ++i) for (int j = 0; j < i; ++j) y[i] -= lu(i, j) * y[j];
        for (int i = N - 1; i >= 0; --i) {
            double s = y[i];
            for (int j = i + 1; j < N; ++j) s -= lu(i, j) * z[j];
            z[i] = s / lu(i, i);
        }
        for (int i = 0; i < N; ++i) inv(colOf[i], col) = z[i];
    }
    return true;
}
inline bool fullPivLuInverse3(const Mat3 &A, Mat3 &inv) { return fullPivLuInverse<3>(A, inv); }

}  // namespace dcreg
