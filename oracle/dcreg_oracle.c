/*
 * dcreg_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).  See dcreg_oracle.h.
 *
 * Build with -ffp-contract=off: the float squared distances that drive the k-NN order and the
 * radius gate must round exactly like the HIP kernel's (which uses explicit non-fused ops).
 */
#define _GNU_SOURCE
#include "dcreg_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ============================================================================================
 * kd-tree: exact k-NN with float32 squared distances (stands for pcl::KdTreeFLANN::nearestKSearch,
 * icp_test_runner.cpp:1722).  Result order: ascending (d2, index).
 * ==========================================================================================*/
#define ORC_LEAF 10

typedef struct {
    int32_t left, right; /* -1 => leaf */
    int32_t begin, end;  /* point range (leaf) */
    int32_t axis;
    float split;
} orc_node;

struct orc_kdtree {
    int64_t n;
    float *pts;    /* [4*n]: x y z, reordered */
    int32_t *idx;  /* original index of reordered point */
    int32_t *inv;  /* reordered position of original point */
    orc_node *nodes;
    int32_t n_nodes, cap_nodes;
};

static int32_t kd_new_node(orc_kdtree *t) {
    if (t->n_nodes == t->cap_nodes) {
        t->cap_nodes = t->cap_nodes ? t->cap_nodes * 2 : 1024;
        t->nodes = (orc_node *)realloc(t->nodes, sizeof(orc_node) * (size_t)t->cap_nodes);
    }
    return t->n_nodes++;
}

/* nth_element on perm[lo,hi) by coordinate axis of src (stride 4 floats), ties by original index */
static inline int kd_less(const float *p, int32_t a, int32_t b, int axis) {
    float va = p[4 * (int64_t)a + axis], vb = p[4 * (int64_t)b + axis];
    return va < vb || (va == vb && a < b);
}
static void kd_select(const float *p, int32_t *perm, int32_t lo, int32_t hi, int32_t nth, int axis) {
    while (hi - lo > 1) {
        int32_t mid = lo + (hi - lo) / 2;
        /* median of three pivot */
        int32_t a = perm[lo], b = perm[mid], c = perm[hi - 1], piv;
        if (kd_less(p, a, b, axis)) {
            piv = kd_less(p, b, c, axis) ? b : (kd_less(p, a, c, axis) ? c : a);
        } else {
            piv = kd_less(p, a, c, axis) ? a : (kd_less(p, b, c, axis) ? c : b);
        }
        int32_t i = lo, j = hi - 1;
        while (i <= j) {
            while (kd_less(p, perm[i], piv, axis)) i++;
            while (kd_less(p, piv, perm[j], axis)) j--;
            if (i <= j) {
                int32_t tmp = perm[i]; perm[i] = perm[j]; perm[j] = tmp;
                i++; j--;
            }
        }
        if (nth <= j) hi = j + 1;
        else if (nth >= i) lo = i;
        else return;
    }
}

static int32_t kd_build_rec(orc_kdtree *t, const float *p, int32_t *perm, int32_t lo, int32_t hi) {
    int32_t id = kd_new_node(t);
    if (hi - lo <= ORC_LEAF) {
        t->nodes[id].left = t->nodes[id].right = -1;
        t->nodes[id].begin = lo; t->nodes[id].end = hi;
        t->nodes[id].axis = 0; t->nodes[id].split = 0.f;
        return id;
    }
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int32_t i = lo; i < hi; ++i)
        for (int a = 0; a < 3; ++a) {
            float v = p[4 * (int64_t)perm[i] + a];
            if (v < mn[a]) mn[a] = v;
            if (v > mx[a]) mx[a] = v;
        }
    int axis = 0;
    if (mx[1] - mn[1] > mx[axis] - mn[axis]) axis = 1;
    if (mx[2] - mn[2] > mx[axis] - mn[axis]) axis = 2;
    int32_t mid = lo + (hi - lo) / 2;
    kd_select(p, perm, lo, hi, mid, axis);
    float split = p[4 * (int64_t)perm[mid] + axis];
    int32_t l = kd_build_rec(t, p, perm, lo, mid);
    int32_t r = kd_build_rec(t, p, perm, mid, hi);
    t->nodes[id].left = l; t->nodes[id].right = r;
    t->nodes[id].begin = lo; t->nodes[id].end = hi;
    t->nodes[id].axis = axis; t->nodes[id].split = split;
    return id;
}

orc_kdtree *orc_kdtree_build(const float *xyz, int64_t n, int64_t stride) {
    orc_kdtree *t = (orc_kdtree *)calloc(1, sizeof(orc_kdtree));
    t->n = n;
    if (n <= 0) return t;
    float *tmp = (float *)malloc(sizeof(float) * 4 * (size_t)n);
    int32_t *perm = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        tmp[4 * i + 0] = xyz[i * stride + 0];
        tmp[4 * i + 1] = xyz[i * stride + 1];
        tmp[4 * i + 2] = xyz[i * stride + 2];
        tmp[4 * i + 3] = 0.f;
        perm[i] = (int32_t)i;
    }
    kd_build_rec(t, tmp, perm, 0, (int32_t)n);
    t->pts = (float *)malloc(sizeof(float) * 4 * (size_t)n);
    t->idx = perm;
    t->inv = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        memcpy(t->pts + 4 * i, tmp + 4 * (int64_t)perm[i], 4 * sizeof(float));
        t->inv[perm[i]] = (int32_t)i;
    }
    free(tmp);
    return t;
}

void orc_kdtree_free(orc_kdtree *t) {
    if (!t) return;
    free(t->pts); free(t->idx); free(t->inv); free(t->nodes); free(t);
}
int64_t orc_kdtree_size(const orc_kdtree *t) { return t ? t->n : 0; }

typedef struct { int k, count; int32_t *idx; float *d2; } kd_best;

static inline void kd_insert(kd_best *b, float d2, int32_t idx) {
    int pos = b->count;
    if (pos == b->k) {
        /* full: reject unless strictly better than the worst in (d2, idx) order */
        float wd = b->d2[pos - 1];
        if (d2 > wd || (d2 == wd && idx > b->idx[pos - 1])) return;
        pos--;
    } else {
        b->count++;
    }
    while (pos > 0 && (b->d2[pos - 1] > d2 || (b->d2[pos - 1] == d2 && b->idx[pos - 1] > idx))) {
        b->d2[pos] = b->d2[pos - 1]; b->idx[pos] = b->idx[pos - 1];
        pos--;
    }
    b->d2[pos] = d2; b->idx[pos] = idx;
}

static void kd_search(const orc_kdtree *t, int32_t id, const float q[3], kd_best *b) {
    const orc_node *nd = &t->nodes[id];
    if (nd->left < 0) {
        for (int32_t i = nd->begin; i < nd->end; ++i) {
            const float *p = t->pts + 4 * (int64_t)i;
            float dx = q[0] - p[0], dy = q[1] - p[1], dz = q[2] - p[2];
            float d2 = dx * dx;      /* FLANN L2_Simple: sequential float accumulation */
            d2 = d2 + dy * dy;
            d2 = d2 + dz * dz;
            kd_insert(b, d2, t->idx[i]);
        }
        return;
    }
    float diff = q[nd->axis] - nd->split;
    int32_t near = diff < 0.f ? nd->left : nd->right;
    int32_t far = diff < 0.f ? nd->right : nd->left;
    kd_search(t, near, q, b);
    /* conservative prune: rounding is monotone, so any far point has d2 >= fl(diff*diff) */
    if (b->count < b->k || diff * diff <= b->d2[b->k - 1]) kd_search(t, far, q, b);
}

int orc_knn(const orc_kdtree *t, const float q[3], int k, int32_t *idx, float *d2) {
    kd_best b = {k, 0, idx, d2};
    if (!t || t->n <= 0 || k <= 0) return 0;
    kd_search(t, 0, q, &b);
    return b.count;
}

void orc_knn_batch(const orc_kdtree *t, const float *q, int64_t n, int64_t stride, int k,
                   int32_t *idx, float *d2, int num_threads) {
#ifdef _OPENMP
    int nt = num_threads > 0 ? num_threads : omp_get_max_threads();
#pragma omp parallel for num_threads(nt) schedule(dynamic, 256)
#endif
    for (int64_t i = 0; i < n; ++i) {
        int found = orc_knn(t, q + i * stride, k, idx + i * k, d2 + i * k);
        for (int j = found; j < k; ++j) { idx[i * k + j] = -1; d2[i * k + j] = INFINITY; }
    }
    (void)num_threads;
}

/* ============================================================================================
 * Eigen::ColPivHouseholderQR restatement (Eigen 3.3.7 ColPivHouseholderQR.h computeInPlace +
 * _solve_impl, Householder.h makeHouseholder / applyHouseholderOnTheLeft).
 * Call sites: icp_test_runner.cpp:1747 (5x3), dcreg.hpp:182,190,197,245,251,255 (6x6).
 * ==========================================================================================*/
#define QR_MAXM 8
#define QR_MAXN 6

int orc_colpiv_qr_solve(int m, int n, const double *A, const double *b, double *x) {
    double qr[QR_MAXM][QR_MAXN], hc[QR_MAXN], cnu[QR_MAXN], cnd[QR_MAXN], c[QR_MAXM];
    int perm[QR_MAXN];
    const double eps = DBL_EPSILON;
    for (int i = 0; i < m; ++i) for (int j = 0; j < n; ++j) qr[i][j] = A[i * n + j];
    for (int j = 0; j < n; ++j) perm[j] = j;
    double maxnorm = 0.0;
    for (int j = 0; j < n; ++j) {
        double s = 0.0;
        for (int i = 0; i < m; ++i) s += qr[i][j] * qr[i][j];
        cnu[j] = cnd[j] = sqrt(s);
        if (cnu[j] > maxnorm) maxnorm = cnu[j];
    }
    const double threshold_helper = (maxnorm * eps) * (maxnorm * eps) / (double)m;
    const double norm_downdate_threshold = sqrt(eps);
    const int size = m < n ? m : n;
    int nonzero_pivots = size;
    for (int k = 0; k < size; ++k) {
        int big = k;
        for (int j = k + 1; j < n; ++j) if (cnu[j] > cnu[big]) big = j;
        double big_sq = cnu[big] * cnu[big];
        if (nonzero_pivots == size && big_sq < threshold_helper * (double)(m - k)) nonzero_pivots = k;
        if (big != k) {
            for (int i = 0; i < m; ++i) { double tmp = qr[i][k]; qr[i][k] = qr[i][big]; qr[i][big] = tmp; }
            double tmp = cnu[k]; cnu[k] = cnu[big]; cnu[big] = tmp;
            tmp = cnd[k]; cnd[k] = cnd[big]; cnd[big] = tmp;
            int ti = perm[k]; perm[k] = perm[big]; perm[big] = ti;
        }
        /* makeHouseholderInPlace on qr[k..m-1][k] */
        double tail_sq = 0.0;
        for (int i = k + 1; i < m; ++i) tail_sq += qr[i][k] * qr[i][k];
        double c0 = qr[k][k], beta, tau;
        if (tail_sq <= DBL_MIN) {
            tau = 0.0; beta = c0;
            for (int i = k + 1; i < m; ++i) qr[i][k] = 0.0;
        } else {
            beta = sqrt(c0 * c0 + tail_sq);
            if (c0 >= 0.0) beta = -beta;
            for (int i = k + 1; i < m; ++i) qr[i][k] = qr[i][k] / (c0 - beta);
            tau = (beta - c0) / beta;
        }
        hc[k] = tau;
        qr[k][k] = beta;
        /* applyHouseholderOnTheLeft to the trailing block */
        if (tau != 0.0) {
            for (int j = k + 1; j < n; ++j) {
                double tmp = 0.0;
                for (int i = k + 1; i < m; ++i) tmp += qr[i][k] * qr[i][j];
                tmp += qr[k][j];
                qr[k][j] -= tau * tmp;
                for (int i = k + 1; i < m; ++i) qr[i][j] -= tau * qr[i][k] * tmp;
            }
        }
        /* LAPACK-style column-norm downdate (lawn176) */
        for (int j = k + 1; j < n; ++j) {
            if (cnu[j] != 0.0) {
                double temp = fabs(qr[k][j]) / cnu[j];
                temp = (1.0 + temp) * (1.0 - temp);
                if (temp < 0.0) temp = 0.0;
                double ratio = cnu[j] / cnd[j];
                double temp2 = temp * ratio * ratio;
                if (temp2 <= norm_downdate_threshold) {
                    double s = 0.0;
                    for (int i = k + 1; i < m; ++i) s += qr[i][j] * qr[i][j];
                    cnd[j] = sqrt(s);
                    cnu[j] = cnd[j];
                } else {
                    cnu[j] *= sqrt(temp);
                }
            }
        }
    }
    /* solve */
    for (int j = 0; j < n; ++j) x[j] = 0.0;
    if (nonzero_pivots == 0) return 0;
    for (int i = 0; i < m; ++i) c[i] = b[i];
    for (int k = 0; k < nonzero_pivots; ++k) { /* c <- H_k c */
        if (hc[k] == 0.0) continue;
        double tmp = c[k];
        for (int i = k + 1; i < m; ++i) tmp += qr[i][k] * c[i];
        c[k] -= hc[k] * tmp;
        for (int i = k + 1; i < m; ++i) c[i] -= hc[k] * qr[i][k] * tmp;
    }
    for (int i = nonzero_pivots - 1; i >= 0; --i) { /* back substitution */
        double s = c[i];
        for (int j = i + 1; j < nonzero_pivots; ++j) s -= qr[i][j] * c[j];
        c[i] = s / qr[i][i];
    }
    for (int i = 0; i < nonzero_pivots; ++i) x[perm[i]] = c[i];
    return nonzero_pivots;
}

/* ============================================================================================
 * Symmetric eigen-decomposition, cyclic Jacobi (stands for Eigen::SelfAdjointEigenSolver,
 * dcreg.hpp:62-66, icp_test_runner.cpp:2426-2449).  Ascending eigenvalues.
 * ==========================================================================================*/
void orc_sym_eig(int n, const double *Ain, double *w, double *Vout) {
    double A[6][6], V[6][6];
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
        A[i][j] = 0.5 * (Ain[i * n + j] + Ain[j * n + i]);
        V[i][j] = (i == j) ? 1.0 : 0.0;
    }
    for (int sweep = 0; sweep < 64; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < n; ++i) {
            diag += A[i][i] * A[i][i];
            for (int j = i + 1; j < n; ++j) off += A[i][j] * A[i][j];
        }
        if (off == 0.0 || off <= 1e-34 * diag) break;
        for (int p = 0; p < n - 1; ++p) for (int q = p + 1; q < n; ++q) {
            double apq = A[p][q];
            if (apq == 0.0) continue;
            double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
            double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
            for (int k = 0; k < n; ++k) {
                double akp = A[k][p], akq = A[k][q];
                A[k][p] = cs * akp - sn * akq;
                A[k][q] = sn * akp + cs * akq;
            }
            for (int k = 0; k < n; ++k) {
                double apk = A[p][k], aqk = A[q][k];
                A[p][k] = cs * apk - sn * aqk;
                A[q][k] = sn * apk + cs * aqk;
            }
            for (int k = 0; k < n; ++k) {
                double vkp = V[k][p], vkq = V[k][q];
                V[k][p] = cs * vkp - sn * vkq;
                V[k][q] = sn * vkp + cs * vkq;
            }
        }
    }
    int order[6];
    for (int i = 0; i < n; ++i) order[i] = i;
    for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j)
        if (A[order[j]][order[j]] < A[order[i]][order[i]]) { int tmp = order[i]; order[i] = order[j]; order[j] = tmp; }
    for (int i = 0; i < n; ++i) {
        w[i] = A[order[i]][order[i]];
        for (int k = 0; k < n; ++k) Vout[k * n + i] = V[k][order[i]];
    }
}

/* Eigen::FullPivLU<MatrixNd>::isInvertible()/inverse() (icp_test_runner.cpp:2422-2445 for the 3x3 Schur blocks,
 * :2016-2018 for the 6x6 covariance).  rank counts pivots with |p| > eps*n*|maxpivot| (FullPivLU::threshold()). */
int orc_inv_fullpiv(int n, const double *Ain, double *Ainv) {
    double lu[6][6];
    int rp[6] = {0, 1, 2, 3, 4, 5}, cp[6] = {0, 1, 2, 3, 4, 5};
    if (n < 1 || n > 6) return 0;
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) lu[i][j] = Ain[i * n + j];
    double maxpivot = 0.0, piv[6] = {0, 0, 0, 0, 0, 0};
    int nonzero = n;
    for (int k = 0; k < n; ++k) {
        int bi = k, bj = k; double best = -1.0;
        for (int i = k; i < n; ++i) for (int j = k; j < n; ++j)
            if (fabs(lu[i][j]) > best) { best = fabs(lu[i][j]); bi = i; bj = j; }
        if (best == 0.0) { nonzero = k; break; }
        if (best > maxpivot) maxpivot = best;
        if (bi != k) { for (int j = 0; j < n; ++j) { double t = lu[k][j]; lu[k][j] = lu[bi][j]; lu[bi][j] = t; } int t = rp[k]; rp[k] = rp[bi]; rp[bi] = t; }
        if (bj != k) { for (int i = 0; i < n; ++i) { double t = lu[i][k]; lu[i][k] = lu[i][bj]; lu[i][bj] = t; } int t = cp[k]; cp[k] = cp[bj]; cp[bj] = t; }
        piv[k] = lu[k][k];
        for (int i = k + 1; i < n; ++i) {
            lu[i][k] /= lu[k][k];
            for (int j = k + 1; j < n; ++j) lu[i][j] -= lu[i][k] * lu[k][j];
        }
    }
    if (nonzero < n) return 0;
    double thr = DBL_EPSILON * (double)n * maxpivot;
    for (int k = 0; k < n; ++k) if (!(fabs(piv[k]) > thr)) return 0;
    /* solve A X = I :  P A Q = L U  =>  A^-1 = Q U^-1 L^-1 P */
    for (int col = 0; col < n; ++col) {
        double y[6], z[6];
        for (int i = 0; i < n; ++i) y[i] = (rp[i] == col) ? 1.0 : 0.0;
        for (int i = 0; i < n; ++i) for (int j = 0; j < i; ++j) y[i] -= lu[i][j] * y[j];
        for (int i = n - 1; i >= 0; --i) {
            double s = y[i];
            for (int j = i + 1; j < n; ++j) s -= lu[i][j] * z[j];
            z[i] = s / lu[i][i];
        }
        for (int i = 0; i < n; ++i) Ainv[cp[i] * n + col] = z[i];
    }
    return 1;
}
int orc_inv3_fullpiv(const double *Ain, double *Ainv) { return orc_inv_fullpiv(3, Ain, Ainv); }

/* ============================================================================================
 * Hot path: one linearisation (icp_test_runner.cpp:1704-1919), SURVEY Appendix A steps 1-8.
 * ==========================================================================================*/
int orc_plane_fit(const double Q[15], double n_out[3], double *d_out, double *ps_out) {
    /* icp_test_runner.cpp:1727-1760 : [q_j] x = -1 (5x3), n = x/|x|, d = 1/|x| */
    const double rhs[5] = {-1.0, -1.0, -1.0, -1.0, -1.0};
    double x[3];
    orc_colpiv_qr_solve(5, 3, Q, rhs, x);
    double ps = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    *ps_out = ps;
    if (!(ps > 0.0)) { n_out[0] = n_out[1] = n_out[2] = 0.0; *d_out = 0.0; return 0; }
    n_out[0] = x[0] / ps; n_out[1] = x[1] / ps; n_out[2] = x[2] / ps;
    *d_out = 1.0 / ps;
    return 1;
}

typedef struct { double a[6]; double b; double r; uint8_t flag; uint8_t has_pt; } orc_row;

static void mat3_mul3(const double *A, const double *B, const double *C3, double *out) {
    double T[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0.0; for (int k = 0; k < 3; ++k) s += A[i * 3 + k] * B[k * 3 + j]; T[i * 3 + j] = s; }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0.0; for (int k = 0; k < 3; ++k) s += T[i * 3 + k] * C3[k * 3 + j]; out[i * 3 + j] = s; }
}

void orc_euler_dR(double roll, double pitch, double yaw, double dR[27]) {
    const double cr = cos(roll), sr = sin(roll), cp = cos(pitch), sp = sin(pitch), cy = cos(yaw), sy = sin(yaw);
    const double Rx[9] = {1, 0, 0, 0, cr, -sr, 0, sr, cr}, dRx[9] = {0, 0, 0, 0, -sr, -cr, 0, cr, -sr};
    const double Ry[9] = {cp, 0, sp, 0, 1, 0, -sp, 0, cp}, dRy[9] = {-sp, 0, cp, 0, 0, 0, -cp, 0, -sp};
    const double Rz[9] = {cy, -sy, 0, sy, cy, 0, 0, 0, 1}, dRz[9] = {-sy, -cy, 0, cy, -sy, 0, 0, 0, 0};
    mat3_mul3(Rz, Ry, dRx, dR); mat3_mul3(Rz, dRy, Rx, dR + 9); mat3_mul3(dRz, Ry, Rx, dR + 18);
}

/* second engine, icp_test_runner.cpp:2299-2346, RESTATED AS WRITTEN (parameterization 1).  The reference names its trigonometric
 * terms after LOAM's camera frame - srx / crx = sin / cos(pitch), sry / cry of yaw, srz / crz of roll (:2299-2304) -, relabels the
 * axes of the body-frame point and of the float-stored weighted normal (x <- y, y <- z, z <- x, :2309-2315) and then multiplies the
 * three brackets of arx by coeff.z, coeff.x, coeff.y (:2323-2326), those of ary by coeff.z, coeff.y (:2327-2330) and those of arz by
 * coeff.z, coeff.x, coeff.y (:2331-2335).  LOAM / LIO-SAM multiply the same brackets by coeff.x, coeff.y, coeff.z: with that order the
 * row is the derivative of c . (Rz(yaw) Ry(pitch) Rx(roll) p) by roll / pitch / yaw (orc_euler_row_exact below); with the reference's
 * cyclic permutation it is NOT (tests/test_euler_engine.py measures the difference).  Bar of this tier = what the reference
 * computes, so this is the default; every product and sum below is taken in the order of the source text (double arithmetic,
 * -ffp-contract=off).  Row order :2338-2343: [arz, arx, ary, coeff.z, coeff.x, coeff.y].  Kept out of line so that the pinned SO(3)
 * row compiles exactly as before (inlining changed that path's rounding in the 16th digit, which ME-TReg's trace amplifies). */
static __attribute__((noinline)) void orc_euler_row(const orc_lin_params *prm, float px, float py, float pz,
                                                    float cx, float cy, float cz, double a[6]) {
    const double srx = sin(prm->euler_rpy[1]), crx = cos(prm->euler_rpy[1]);   /* :2299-2300 (pitch) */
    const double sry = sin(prm->euler_rpy[2]), cry = cos(prm->euler_rpy[2]);   /* :2301-2302 (yaw) */
    const double srz = sin(prm->euler_rpy[0]), crz = cos(prm->euler_rpy[0]);   /* :2303-2304 (roll) */
    struct { float x, y, z; } pointOri, coeff;
    pointOri.x = py; pointOri.y = pz; pointOri.z = px;                          /* :2309-2311 */
    coeff.x = cy; coeff.y = cz; coeff.z = cx;                                   /* :2313-2315 */
    const double crx_sry = crx * sry, crz_sry = crz * sry, srx_sry = srx * sry, srx_srz = srx * srz;   /* :2319-2322 */
    const double arx = (crx_sry * srz * pointOri.x + crx * crz_sry * pointOri.y - srx_sry * pointOri.z) * coeff.z +
                       (-srx_srz * pointOri.x - crz * srx * pointOri.y - crx * pointOri.z) * coeff.x +
                       (crx * cry * srz * pointOri.x + crx * cry * crz * pointOri.y - cry * srx * pointOri.z) * coeff.y;
    const double ary = ((cry * srx_srz - crz_sry) * pointOri.x + (sry * srz + cry * crz * srx) * pointOri.y +
                        crx * cry * pointOri.z) * coeff.z +
                       ((-cry * crz - srx_sry * srz) * pointOri.x + (cry * srz - crz * srx_sry) * pointOri.y -
                        crx_sry * pointOri.z) * coeff.y;
    const double arz = ((crz * srx_sry - cry * srz) * pointOri.x + (-cry * crz - srx_sry * srz) * pointOri.y) * coeff.z +
                       (crx * crz * pointOri.x - crx * srz * pointOri.y) * coeff.x +
                       ((sry * srz + cry * crz * srx) * pointOri.x + (crz_sry - cry * srx_srz) * pointOri.y) * coeff.y;
    a[0] = arz; a[1] = arx; a[2] = ary;                                         /* :2338-2340 */
    a[3] = coeff.z; a[4] = coeff.x; a[5] = coeff.y;                             /* :2341-2343 */
}

/* parameterization 2 (additive, not in the reference): the exact derivative of c . (Rz(yaw) Ry(pitch) Rx(roll) p) by roll / pitch /
 * yaw = LOAM's / LIO-SAM's row with its x<-y, y<-z, z<-x relabelling undone; translation columns = c. */
static __attribute__((noinline)) void orc_euler_row_exact(const orc_lin_params *prm, double px, double py, double pz,
                                                          float cx, float cy, float cz, double a[6]) {
    double dR[27];
    orc_euler_dR(prm->euler_rpy[0], prm->euler_rpy[1], prm->euler_rpy[2], dR);
    const double c[3] = {(double)cx, (double)cy, (double)cz};
    for (int k = 0; k < 3; ++k) {
        const double *D = dR + 9 * k;
        a[k] = c[0] * (D[0] * px + D[1] * py + D[2] * pz) + c[1] * (D[3] * px + D[4] * py + D[5] * pz) +
               c[2] * (D[6] * px + D[7] * py + D[8] * pz);
        a[3 + k] = c[k];
    }
}

/* both rows for one point and one weighted normal (tests: the restatement against the reference's text, and the permutation) */
void orc_euler_rows(const double rpy[3], const float p[3], const float c[3], double literal[6], double exact[6]) {
    orc_lin_params prm;
    memset(&prm, 0, sizeof(prm));
    prm.euler_rpy[0] = rpy[0]; prm.euler_rpy[1] = rpy[1]; prm.euler_rpy[2] = rpy[2];
    orc_euler_row(&prm, p[0], p[1], p[2], c[0], c[1], c[2], literal);
    orc_euler_row_exact(&prm, (double)p[0], (double)p[1], (double)p[2], c[0], c[1], c[2], exact);
}

static void orc_point_row(const orc_kdtree *tree, const float *P, const double R[9], const double t[3],
                          const orc_lin_params *prm, orc_row *row, orc_lin_debug *dbg, int64_t i) {
    row->flag = 0; row->has_pt = 0; row->b = 0.0; row->r = 0.0;
    for (int k = 0; k < 6; ++k) row->a[k] = 0.0;
    const double px = P[0], py = P[1], pz = P[2];
    /* utils.hpp:630-636: double transform, float store */
    float q[3];
    q[0] = (float)(R[0] * px + R[1] * py + R[2] * pz + t[0]);
    q[1] = (float)(R[3] * px + R[4] * py + R[5] * pz + t[1]);
    q[2] = (float)(R[6] * px + R[7] * py + R[8] * pz + t[2]);
    int32_t idx[5]; float d2[5];
    int found = orc_knn(tree, q, 5, idx, d2);
    if (dbg && dbg->nn_idx) for (int j = 0; j < 5; ++j) dbg->nn_idx[5 * i + j] = j < found ? idx[j] : -1;
    if (dbg && dbg->nn_d2) for (int j = 0; j < 5; ++j) dbg->nn_d2[5 * i + j] = j < found ? d2[j] : INFINITY;
    const double R2 = prm->search_radius * prm->search_radius;
    if (!(found == 5 && (double)d2[4] < R2)) return;             /* :1726 */
    row->has_pt = 1;                                             /* :1731 */
    double Q[15];
    for (int j = 0; j < 5; ++j) { /* :1738 neighbour coordinates, float -> double */
        const float *p = tree->pts + 4 * (int64_t)tree->inv[idx[j]];
        Q[3 * j + 0] = p[0]; Q[3 * j + 1] = p[1]; Q[3 * j + 2] = p[2];
    }
    double n[3], d, ps;
    orc_plane_fit(Q, n, &d, &ps);
    if (ps < prm->min_normal_norm) { row->flag = 2; return; }    /* :1752 */
    double maxd = 0.0;                                           /* :1763-1770 */
    for (int j = 0; j < 5; ++j) {
        double dist = n[0] * Q[3 * j] + n[1] * Q[3 * j + 1] + n[2] * Q[3 * j + 2] + d;
        dist *= dist;
        if (dist > maxd) maxd = dist;
    }
    if (!(maxd < prm->max_plane_thickness_sq)) { row->flag = 3; return; }   /* :1773 */
    double r = n[0] * (double)q[0] + n[1] * (double)q[1] + n[2] * (double)q[2] + d; /* :1774 */
    double absr = fabs(r);
    double s = 1.0 - prm->weight_slope * absr;                   /* :1776 */
    if (s < 0.0) s = 0.0;
    double ds = 0.0;
    if (prm->use_weight_derivative && s > 0.0 && s < 1.0)        /* :1780-1783 */
        ds = -prm->weight_slope * (r > 0.0 ? 1.0 : -1.0);
    if (dbg) {
        if (dbg->normal) { dbg->normal[3 * i] = n[0]; dbg->normal[3 * i + 1] = n[1]; dbg->normal[3 * i + 2] = n[2]; }
        if (dbg->r) dbg->r[i] = r;
        if (dbg->s) dbg->s[i] = s;
    }
    if (!(s > prm->weight_min)) { row->flag = 4; return; }       /* :1785 */
    /* :1786-1790 float stores */
    float cx = (float)(s * n[0]), cy = (float)(s * n[1]), cz = (float)(s * n[2]), ci = (float)(s * r);
    /* :1889 unweighted normal recovered from the float coefficients */
    double nx = (double)cx / s, ny = (double)cy / s, nz = (double)cz / s;
    /* math_utils.hpp:102-121 : J = [ -n^T R [p]x , n^T R ] = [ (p x m)^T , m^T ], m = R^T n */
    double m0 = R[0] * nx + R[3] * ny + R[6] * nz;
    double m1 = R[1] * nx + R[4] * ny + R[7] * nz;
    double m2 = R[2] * nx + R[5] * ny + R[8] * nz;
    double c0 = py * m2 - pz * m1, c1 = pz * m0 - px * m2, c2 = px * m1 - py * m0;
    double w = s + r * ds;                                       /* :1898 */
    row->a[0] = w * c0; row->a[1] = w * c1; row->a[2] = w * c2;
    row->a[3] = w * m0; row->a[4] = w * m1; row->a[5] = w * m2;
    if (prm->parameterization == 1) orc_euler_row(prm, P[0], P[1], P[2], cx, cy, cz, row->a);
    else if (prm->parameterization == 2) orc_euler_row_exact(prm, px, py, pz, cx, cy, cz, row->a);
    row->b = -(double)ci;                                        /* :1906 */
    row->r = r;
    row->flag = 1;
}

#define ORC_CHUNK 256

int orc_linearize(const orc_kdtree *tree, const float *src, int64_t n_src, int64_t stride,
                  const double R[9], const double t[3], const orc_lin_params *prm,
                  orc_lin_out *out, orc_lin_debug *dbg) {
    memset(out, 0, sizeof(*out));
    if (!tree || n_src < 0) return -1;
    int64_t n_chunks = (n_src + ORC_CHUNK - 1) / ORC_CHUNK;
    double *part = (double *)calloc((size_t)(n_chunks > 0 ? n_chunks : 1) * 32, sizeof(double));
#ifdef _OPENMP
    int nt = prm->num_threads > 0 ? prm->num_threads : omp_get_max_threads();
#pragma omp parallel for num_threads(nt) schedule(dynamic, 1)
#endif
    for (int64_t c = 0; c < n_chunks; ++c) {
        double acc[32];
        for (int k = 0; k < 32; ++k) acc[k] = 0.0;
        int64_t lo = c * ORC_CHUNK, hi = lo + ORC_CHUNK < n_src ? lo + ORC_CHUNK : n_src;
        for (int64_t i = lo; i < hi; ++i) {
            orc_row row;
            orc_point_row(tree, src + i * stride, R, t, prm, &row, dbg, i);
            if (dbg && dbg->flag) dbg->flag[i] = row.flag;
            if (row.has_pt) acc[30] += 1.0;
            if (row.flag != 1) continue;
            int idx = 0;
            for (int j = 0; j < 6; ++j) for (int k = j; k < 6; ++k) acc[idx++] += row.a[j] * row.a[k];
            for (int j = 0; j < 6; ++j) acc[21 + j] += row.a[j] * row.b;
            acc[27] += row.r * row.r;
            acc[28] += row.b * row.b;
            acc[29] += 1.0;
        }
        memcpy(part + 32 * c, acc, sizeof(acc));
    }
    double tot[32];
    for (int k = 0; k < 32; ++k) tot[k] = 0.0;
    for (int64_t c = 0; c < n_chunks; ++c) for (int k = 0; k < 32; ++k) tot[k] += part[32 * c + k];
    free(part);
    memcpy(out->H_upper, tot, 21 * sizeof(double));
    memcpy(out->g, tot + 21, 6 * sizeof(double));
    out->sum_r2 = tot[27]; out->sum_b2 = tot[28];
    out->n_eff = (int64_t)tot[29]; out->n_pt = (int64_t)tot[30];
    return 0;
}

/* ============================================================================================
 * 6x6 analysis and solvers (dcreg.hpp:45-264) + Schur block (icp_test_runner.cpp:2418-2469)
 * + the unreleased SCHUR_CONDITION_NUMBER / PRECONDITIONED_CG pieces.
 * ==========================================================================================*/
void orc_unpack_H(const double U[21], double H[36]) {
    int idx = 0;
    for (int i = 0; i < 6; ++i) for (int j = i; j < 6; ++j) { H[i * 6 + j] = U[idx]; H[j * 6 + i] = U[idx]; idx++; }
}

static void mat3_mul(const double *A, const double *B, double *C) {
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        double s = 0.0;
        for (int k = 0; k < 3; ++k) s += A[i * 3 + k] * B[k * 3 + j];
        C[i * 3 + j] = s;
    }
}

static double minc(const double *v, int n) { double m = v[0]; for (int i = 1; i < n; ++i) if (v[i] < m) m = v[i]; return m; }
static double maxc(const double *v, int n) { double m = v[0]; for (int i = 1; i < n; ++i) if (v[i] > m) m = v[i]; return m; }

static void orc_schur(const double H[36], const orc_config *cfg, orc_analysis *r) {
    double Hrr[9], Htt[9], Hrt[9], Htr[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        Hrr[i * 3 + j] = H[i * 6 + j];
        Htt[i * 3 + j] = H[(i + 3) * 6 + j + 3];
        Hrt[i * 3 + j] = H[i * 6 + j + 3];
        Htr[i * 3 + j] = H[(i + 3) * 6 + j];
    }
    double V[9];
    orc_sym_eig(3, Hrr, r->lambda_sub_rot, V);      /* :2426-2432 */
    r->cond_diag_rot = maxc(r->lambda_sub_rot, 3) / fmax(minc(r->lambda_sub_rot, 3), 1e-12);
    orc_sym_eig(3, Htt, r->lambda_sub_trans, V);    /* :2433-2439 */
    r->cond_diag_trans = maxc(r->lambda_sub_trans, 3) / fmax(minc(r->lambda_sub_trans, 3), 1e-12);
    double Hrr_inv[9], Htt_inv[9];
    int ok = orc_inv3_fullpiv(Htt, Htt_inv) & orc_inv3_fullpiv(Hrr, Hrr_inv);   /* :2443 */
    for (int i = 0; i < 36; ++i) r->P_preconditioner[i] = (i % 7 == 0) ? 1.0 : 0.0;
    if (!ok) {
        r->cond_schur_rot = r->cond_schur_trans = INFINITY;           /* :2468 */
        return;
    }
    double T1[9], T2[9], SR[9], ST[9];
    mat3_mul(Hrt, Htt_inv, T1); mat3_mul(T1, Htr, T2);
    for (int i = 0; i < 9; ++i) SR[i] = Hrr[i] - T2[i];               /* :2446 */
    mat3_mul(Htr, Hrr_inv, T1); mat3_mul(T1, Hrt, T2);
    for (int i = 0; i < 9; ++i) ST[i] = Htt[i] - T2[i];               /* :2447 */
    orc_sym_eig(3, SR, r->lambda_schur_rot, r->schur_V_rot);
    orc_sym_eig(3, ST, r->lambda_schur_trans, r->schur_V_trans);
    r->cond_schur_rot = maxc(r->lambda_schur_rot, 3) / fmax(minc(r->lambda_schur_rot, 3), 1e-12);       /* :2456 */
    r->cond_schur_trans = maxc(r->lambda_schur_trans, 3) / fmax(minc(r->lambda_schur_trans, 3), 1e-12); /* :2458 */
    /* eigenvalue-clamped block preconditioner */
    for (int blk = 0; blk < 2; ++blk) {
        const double *lam = blk ? r->lambda_schur_trans : r->lambda_schur_rot;
        const double *Vb = blk ? r->schur_V_trans : r->schur_V_rot;
        double lmax = maxc(lam, 3), floor_l = lmax / cfg->kappa_target;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) {
                double lt = lam[k] > floor_l ? lam[k] : floor_l;
                s += Vb[i * 3 + k] * Vb[j * 3 + k] / lt;
            }
            r->P_preconditioner[(i + 3 * blk) * 6 + (j + 3 * blk)] = s;
        }
    }
}

void orc_analyze(const double H[36], int detection, int handling, const orc_config *cfg, orc_analysis *r) {
    memset(r, 0, sizeof(*r));
    r->cond_schur_rot = r->cond_schur_trans = r->cond_diag_rot = r->cond_diag_trans = NAN;
    for (int i = 0; i < 3; ++i)
        r->lambda_schur_rot[i] = r->lambda_schur_trans[i] = r->lambda_sub_rot[i] = r->lambda_sub_trans[i] = NAN;
    for (int i = 0; i < 36; ++i) r->P_preconditioner[i] = (i % 7 == 0) ? 1.0 : 0.0;
    for (int i = 0; i < 9; ++i) r->schur_V_rot[i] = r->schur_V_trans[i] = (i % 4 == 0) ? 1.0 : 0.0;

    /* dcreg.hpp:66-80 */
    orc_sym_eig(6, H, r->eigenvalues_full, r->eigenvectors_full);
    const double *ev = r->eigenvalues_full;
    r->cond_full_sub_trans = fabs(ev[2]) / fmax(fabs(ev[0]), 1e-12);
    r->cond_full_sub_rot = fabs(ev[5]) / fmax(fabs(ev[3]), 1e-12);
    /* dcreg.hpp:83-89 : singular values of a symmetric matrix = |eigenvalues|, descending */
    double sv[6];
    for (int i = 0; i < 6; ++i) sv[i] = fabs(ev[i]);
    for (int i = 0; i < 6; ++i) for (int j = i + 1; j < 6; ++j) if (sv[j] > sv[i]) { double t = sv[i]; sv[i] = sv[j]; sv[j] = t; }
    memcpy(r->singular_values, sv, sizeof(sv));
    r->cond_full = sv[5] > 1e-12 ? sv[0] / sv[5] : INFINITY;

    if (detection == ORC_DET_SCHUR_CONDITION_NUMBER || handling == ORC_HAND_PRECONDITIONED_CG ||
        cfg->always_compute_schur)
        orc_schur(H, cfg, r);

    switch (detection) {
    case ORC_DET_SCHUR_CONDITION_NUMBER: { /* unreleased; SURVEY App. C.2 */
        if (isfinite(r->cond_schur_rot) && isfinite(r->cond_schur_trans)) {
            double lr = maxc(r->lambda_schur_rot, 3), lt = maxc(r->lambda_schur_trans, 3);
            for (int i = 0; i < 3; ++i) {
                if (lr / fmax(r->lambda_schur_rot[i], 1e-12) > cfg->thres_cond) { r->mask[i] = 1; r->is_degenerate = 1; }
                if (lt / fmax(r->lambda_schur_trans[i], 1e-12) > cfg->thres_cond) { r->mask[3 + i] = 1; r->is_degenerate = 1; }
            }
        } else {
            r->is_degenerate = 1;
            for (int i = 0; i < 6; ++i) r->mask[i] = 1;
        }
        break;
    }
    case ORC_DET_FULL_EVD_MIN_EIGENVALUE: /* dcreg.hpp:100-110 */
        for (int i = 0; i < 6; ++i) if (ev[i] < cfg->thres_eig) { r->is_degenerate = 1; r->mask[i] = 1; }
        break;
    case ORC_DET_EVD_SUB_CONDITION: /* dcreg.hpp:112-126 (cond_diag_* are NaN unless Schur block ran) */
        r->is_degenerate = (r->cond_diag_rot > cfg->thres_cond || r->cond_diag_trans > cfg->thres_cond);
        if (r->is_degenerate) {
            if (r->cond_diag_trans > cfg->thres_cond) for (int i = 0; i < 3; ++i) r->mask[i + 3] = 1;
            if (r->cond_diag_rot > cfg->thres_cond) for (int i = 0; i < 3; ++i) r->mask[i] = 1;
        }
        break;
    case ORC_DET_FULL_SVD_CONDITION: /* dcreg.hpp:128-153 */
        r->is_degenerate = r->cond_full > cfg->thres_cond;
        if (r->is_degenerate) {
            double mx = maxc(ev, 6);
            for (int i = 0; i < 6; ++i) if (mx / ev[i] > cfg->thres_cond) r->mask[i] = 1;
        }
        break;
    default: break;
    }
}

void orc_pcg(const double A[36], const double b[6], const double P[36], int max_iter, double tol,
             double x[6], int *iters) {
    double r[6], z[6], p[6], Ap[6];
    double bnorm = 0.0;
    for (int i = 0; i < 6; ++i) { x[i] = 0.0; r[i] = b[i]; bnorm += b[i] * b[i]; }
    bnorm = sqrt(bnorm);
    *iters = 0;
    if (bnorm == 0.0) return;
    for (int i = 0; i < 6; ++i) { z[i] = 0.0; for (int j = 0; j < 6; ++j) z[i] += P[i * 6 + j] * r[j]; p[i] = z[i]; }
    double rz = 0.0;
    for (int i = 0; i < 6; ++i) rz += r[i] * z[i];
    for (int k = 0; k < max_iter; ++k) {
        double pAp = 0.0;
        for (int i = 0; i < 6; ++i) { Ap[i] = 0.0; for (int j = 0; j < 6; ++j) Ap[i] += A[i * 6 + j] * p[j]; pAp += p[i] * Ap[i]; }
        if (!(pAp > 0.0)) break;
        double alpha = rz / pAp, rn = 0.0;
        for (int i = 0; i < 6; ++i) { x[i] += alpha * p[i]; r[i] -= alpha * Ap[i]; rn += r[i] * r[i]; }
        *iters = k + 1;
        if (sqrt(rn) <= tol * bnorm) break;
        double rz_new = 0.0;
        for (int i = 0; i < 6; ++i) { z[i] = 0.0; for (int j = 0; j < 6; ++j) z[i] += P[i * 6 + j] * r[j]; rz_new += r[i] * z[i]; }
        double beta = rz_new / rz;
        rz = rz_new;
        for (int i = 0; i < 6; ++i) p[i] = z[i] + beta * p[i];
    }
}

void orc_solve(const double H[36], const double g[6], int handling, const orc_config *cfg,
               orc_analysis *an, double x[6]) {
    switch (handling) {
    case ORC_HAND_STANDARD_REGULARIZATION: { /* dcreg.hpp:177-184 */
        double Hr[36];
        memcpy(Hr, H, sizeof(Hr));
        if (an->is_degenerate) for (int i = 0; i < 6; ++i) Hr[i * 7] += cfg->std_reg_gamma;
        orc_colpiv_qr_solve(6, 6, Hr, g, x);
        break;
    }
    case ORC_HAND_PRECONDITIONED_CG: /* dcreg.hpp:186-193 + App. C.4 */
        if (an->is_degenerate) orc_pcg(H, g, an->P_preconditioner, cfg->pcg_max_iter, cfg->pcg_tolerance, x, &an->pcg_iterations);
        else orc_colpiv_qr_solve(6, 6, H, g, x);
        break;
    case ORC_HAND_SOLUTION_REMAPPING: { /* dcreg.hpp:195-221 */
        orc_colpiv_qr_solve(6, 6, H, g, x);
        if (an->is_degenerate) {
            double y[6] = {0, 0, 0, 0, 0, 0};
            int good = 0;
            for (int i = 0; i < 6; ++i) {
                if (an->mask[i]) continue;
                good++;
                double dot = 0.0;
                for (int k = 0; k < 6; ++k) dot += an->eigenvectors_full[k * 6 + i] * x[k];
                for (int k = 0; k < 6; ++k) y[k] += an->eigenvectors_full[k * 6 + i] * dot;
            }
            for (int k = 0; k < 6; ++k) x[k] = good > 0 ? y[k] : 0.0;
        }
        break;
    }
    case ORC_HAND_TRUNCATED_SVD: { /* dcreg.hpp:223-248: mask is in ascending-EVD order, sigma descending */
        double y[6] = {0, 0, 0, 0, 0, 0};
        int kept = 0;
        for (int i = 0; i < 6; ++i) {
            int e = 5 - i; /* i-th largest singular value <- eigen index e (|lambda| order == lambda order for PSD) */
            /* guard against tiny negative eigenvalues reordering: pick by |lambda| rank */
            double sv = an->singular_values[i];
            if (an->mask[i] || !(sv > 1e-9)) continue;
            /* find eigen index whose |lambda| equals sv (i-th in descending |lambda|) */
            int order[6] = {0, 1, 2, 3, 4, 5};
            for (int a = 0; a < 6; ++a) for (int b2 = a + 1; b2 < 6; ++b2)
                if (fabs(an->eigenvalues_full[order[b2]]) > fabs(an->eigenvalues_full[order[a]])) { int tmp = order[a]; order[a] = order[b2]; order[b2] = tmp; }
            e = order[i];
            double sgn = an->eigenvalues_full[e] < 0.0 ? -1.0 : 1.0;
            double dot = 0.0;
            for (int k = 0; k < 6; ++k) dot += sgn * an->eigenvectors_full[k * 6 + e] * g[k];
            for (int k = 0; k < 6; ++k) y[k] += an->eigenvectors_full[k * 6 + e] * dot / sv;
            kept++;
        }
        for (int k = 0; k < 6; ++k) x[k] = kept ? y[k] : 0.0;
        break;
    }
    default: /* NONE_HAND, ADAPTIVE_REGULARIZATION (no case in the reference -> default) : dcreg.hpp:250-257 */
        orc_colpiv_qr_solve(6, 6, H, g, x);
        break;
    }
}

/* ============================================================================================
 * SE(3) helpers
 * ==========================================================================================*/
void orc_so3_exp(const double w[3], double R[9]) { /* math_utils.hpp:20-33 */
    double th = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    double K[9];
    if (th < 1e-10) {
        K[0] = 0; K[1] = -w[2]; K[2] = w[1]; K[3] = w[2]; K[4] = 0; K[5] = -w[0]; K[6] = -w[1]; K[7] = w[0]; K[8] = 0;
        for (int i = 0; i < 9; ++i) R[i] = K[i] + ((i % 4 == 0) ? 1.0 : 0.0);
        return;
    }
    double a[3] = {w[0] / th, w[1] / th, w[2] / th};
    K[0] = 0; K[1] = -a[2]; K[2] = a[1]; K[3] = a[2]; K[4] = 0; K[5] = -a[0]; K[6] = -a[1]; K[7] = a[0]; K[8] = 0;
    double KK[9];
    mat3_mul(K, K, KK);
    double sn = sin(th), cs = 1.0 - cos(th);
    for (int i = 0; i < 9; ++i) R[i] = ((i % 4 == 0) ? 1.0 : 0.0) + sn * K[i] + cs * KK[i];
}

void orc_boxplus(const double R[9], const double t[3], const double dx[6], double R2[9], double t2[3]) {
    double E[9], Rn[9], tn[3]; /* math_utils.hpp:158-166 */
    orc_so3_exp(dx, E);
    mat3_mul(R, E, Rn);
    for (int i = 0; i < 3; ++i) tn[i] = t[i] + R[i * 3] * dx[3] + R[i * 3 + 1] * dx[4] + R[i * 3 + 2] * dx[5];
    memcpy(R2, Rn, sizeof(Rn)); memcpy(t2, tn, sizeof(tn));
}

void orc_pose6d_to_matrix(double roll, double pitch, double yaw, double x, double y, double z, double T[16]) {
    /* utils.hpp:452-460 : T * Rz(yaw) * Ry(pitch) * Rx(roll) */
    double cr = cos(roll), sr = sin(roll), cp = cos(pitch), sp = sin(pitch), cy = cos(yaw), sy = sin(yaw);
    double Rx[9] = {1, 0, 0, 0, cr, -sr, 0, sr, cr};
    double Ry[9] = {cp, 0, sp, 0, 1, 0, -sp, 0, cp};
    double Rz[9] = {cy, -sy, 0, sy, cy, 0, 0, 0, 1};
    double A[9], B[9];
    mat3_mul(Rz, Ry, A); mat3_mul(A, Rx, B);
    for (int i = 0; i < 16; ++i) T[i] = 0.0;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T[i * 4 + j] = B[i * 3 + j];
    T[3] = x; T[7] = y; T[11] = z; T[15] = 1.0;
}

void orc_pose_error(const double gt[16], const double T[16], double *trans, double *rot_deg) {
    /* utils.hpp:497-535 : E = gt^-1 T ; |t_E| ; AngleAxis(R_E).angle() (via quaternion) */
    double Rg[9], tg[3], Rt[9], tt[3], E[9], te[3];
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) { Rg[i * 3 + j] = gt[i * 4 + j]; Rt[i * 3 + j] = T[i * 4 + j]; } tg[i] = gt[i * 4 + 3]; tt[i] = T[i * 4 + 3]; }
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) { double s = 0.0; for (int k = 0; k < 3; ++k) s += Rg[k * 3 + i] * Rt[k * 3 + j]; E[i * 3 + j] = s; }
        double s = 0.0; for (int k = 0; k < 3; ++k) s += Rg[k * 3 + i] * (tt[k] - tg[k]); te[i] = s;
    }
    *trans = sqrt(te[0] * te[0] + te[1] * te[1] + te[2] * te[2]);
    /* Eigen Quaternion(Matrix3) then AngleAxis(Quaternion) */
    double qw, qx, qy, qz, tr = E[0] + E[4] + E[8];
    if (tr > 0.0) {
        double s = sqrt(tr + 1.0); qw = 0.5 * s; s = 0.5 / s;
        qx = (E[7] - E[5]) * s; qy = (E[2] - E[6]) * s; qz = (E[3] - E[1]) * s;
    } else {
        int i = 0; if (E[4] > E[0]) i = 1; if (E[8] > E[i * 4]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        double s = sqrt(E[i * 4] - E[j * 4] - E[k * 4] + 1.0);
        double q[3]; q[i] = 0.5 * s; s = 0.5 / s;
        qw = (E[k * 3 + j] - E[j * 3 + k]) * s;
        q[j] = (E[j * 3 + i] + E[i * 3 + j]) * s; q[k] = (E[k * 3 + i] + E[i * 3 + k]) * s;
        qx = q[0]; qy = q[1]; qz = q[2];
    }
    double nrm = sqrt(qx * qx + qy * qy + qz * qz);
    double ang = nrm != 0.0 ? 2.0 * atan2(nrm, fabs(qw)) : 0.0;
    *rot_deg = fabs(ang) * 180.0 / M_PI;
}

/* ============================================================================================
 * Engine loop (icp_test_runner.cpp:1694-2037)
 * ==========================================================================================*/
void orc_default_config(orc_config *c) {
    memset(c, 0, sizeof(*c));
    c->search_radius = 1.0; c->max_iterations = 30;
    c->thresh_rot = 1e-5; c->thresh_trans = 1e-3;           /* utils.hpp:139-140 */
    c->thres_cond = 10.0; c->thres_eig = 120.0;             /* utils.hpp:83-84 */
    c->kappa_target = 1.0; c->pcg_tolerance = 1e-6; c->pcg_max_iter = 10;
    c->std_reg_gamma = 0.01; c->adaptive_reg_alpha = 10.0;
    c->use_weight_derivative = 0; c->always_compute_schur = 0; c->num_threads = 0;
    for (int i = 0; i < 16; ++i) c->gt[i] = (i % 5 == 0) ? 1.0 : 0.0;
}

int orc_icp_run(const orc_kdtree *tree, const float *src, int64_t n_src, int64_t stride,
                const double R0[9], const double t0[3], int detection, int handling,
                const orc_config *cfg, orc_iter_log *log, int log_cap, orc_icp_result *res) {
    double R[9], t[3], Hlast[36];
    memcpy(R, R0, sizeof(R)); memcpy(t, t0, sizeof(t));
    for (int i = 0; i < 36; ++i) Hlast[i] = (i % 7 == 0) ? 1.0 : 0.0;
    memset(res, 0, sizeof(*res));
    orc_lin_params prm = {cfg->search_radius, 0.2 * 0.2, 1e-6, 0.9, 0.1, cfg->use_weight_derivative, cfg->num_threads, 0, 0, {0.0, 0.0, 0.0}};
    if (!tree || orc_kdtree_size(tree) == 0 || n_src <= 0) { res->status = 3; return 0; } /* :1635-1646 */
    for (int it = 0; it < cfg->max_iterations; ++it) {
        orc_lin_out lo;
        orc_linearize(tree, src, n_src, stride, R, t, &prm, &lo, NULL);
        if (lo.n_eff < 10) {                                   /* :1847-1854 */
            res->iterations = it + 1; res->converged = 0; res->status = 1;
            memcpy(res->R, R, sizeof(R)); memcpy(res->t, t, sizeof(t));
            goto cov;
        }
        double H[36], x[6];
        orc_unpack_H(lo.H_upper, H);
        orc_analysis an;
        orc_analyze(H, detection, handling, cfg, &an);
        orc_solve(H, lo.g, handling, cfg, &an, x);
        int finite = 1;
        for (int i = 0; i < 6; ++i) if (!isfinite(x[i])) finite = 0;
        if (!finite) {                                          /* :1942-1950 */
            res->iterations = it; res->converged = 0; res->status = 2;
            memcpy(res->R, R, sizeof(R)); memcpy(res->t, t, sizeof(t));
            goto cov;
        }
        orc_boxplus(R, t, x, R, t);                             /* :1953 */
        memcpy(Hlast, H, sizeof(H));
        if (log && it < log_cap) {
            orc_iter_log *L = &log[it];
            memset(L, 0, sizeof(*L));
            L->iter = it; L->n_eff = lo.n_eff; L->n_pt = lo.n_pt;
            L->fitness = (double)lo.n_pt / (double)n_src;       /* :1856 */
            L->rmse = sqrt(lo.sum_r2 / (double)lo.n_eff);       /* :1858 */
            L->objective = 0.5 * lo.sum_b2;                     /* :1919 */
            for (int i = 0; i < 6; ++i) { L->gradient[i] = -lo.g[i]; L->dx[i] = x[i]; }
            for (int i = 0; i < 16; ++i) L->T[i] = (i % 5 == 0) ? 1.0 : 0.0;
            for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) L->T[i * 4 + j] = R[i * 3 + j]; L->T[i * 4 + 3] = t[i]; }
            orc_pose_error(cfg->gt, L->T, &L->trans_err, &L->rot_err_deg);
            memcpy(L->H_upper, lo.H_upper, sizeof(lo.H_upper));
            L->an = an;
        }
        res->iterations = it + 1;
        double dr = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
        double dt = sqrt(x[3] * x[3] + x[4] * x[4] + x[5] * x[5]);
        if (dr < cfg->thresh_rot && dt < cfg->thresh_trans) { res->converged = 1; break; }   /* :1998 */
    }
    memcpy(res->R, R, sizeof(R)); memcpy(res->t, t, sizeof(t));
cov:
    /* :2014-2037 covariance = FullPivLU(H_last).inverse() if converged and invertible, eigenvalue-clamped to 1e-9 when the
     * smallest eigenvalue of the inverse is <= 1e-12 (:2020-2029); otherwise 1e6 I */
    for (int i = 0; i < 36; ++i) res->cov[i] = (i % 7 == 0) ? 1e6 : 0.0;
    if (res->converged) {
        double inv[36];
        if (orc_inv_fullpiv(6, Hlast, inv)) {
            double sym[36], w[6], V[36];
            for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) sym[i * 6 + j] = inv[(i > j ? i : j) * 6 + (i > j ? j : i)];   /* lower triangle */
            orc_sym_eig(6, sym, w, V);
            double mn = w[0];
            for (int i = 1; i < 6; ++i) if (w[i] < mn) mn = w[i];
            if (mn <= 1e-12) {
                for (int i = 0; i < 6; ++i) if (w[i] < 1e-9) w[i] = 1e-9;
                for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) {
                    double s2 = 0.0;
                    for (int k = 0; k < 6; ++k) s2 += V[i * 6 + k] * w[k] * V[j * 6 + k];
                    inv[i * 6 + j] = s2;
                }
            }
            memcpy(res->cov, inv, sizeof(inv));
        }
    }
    return res->converged;
}

/* second engine (icp_test_runner.cpp:2064-2830) */
int orc_icp_run_euler(const orc_kdtree *tree, const float *src, int64_t n_src, int64_t stride,
                      const double pose6d[6], int detection, int handling, const orc_config *cfg,
                      orc_iter_log *log, int log_cap, orc_icp_result *res, double final_pose6d[6]) {
    double pose[6], T[16], R[9], t[3];
    memcpy(pose, pose6d, sizeof(pose));
    memset(res, 0, sizeof(*res));
    for (int i = 0; i < 36; ++i) res->cov[i] = (i % 7 == 0) ? 1e6 : 0.0;
    orc_lin_params prm = {cfg->search_radius, 0.2 * 0.2, 1e-6, 0.9, 0.1, 0, cfg->num_threads, cfg->euler_exact_jacobian ? 2 : 1, 0, {0.0, 0.0, 0.0}};
    if (!tree || orc_kdtree_size(tree) == 0 || n_src <= 0) { res->status = 3; return 0; }
    double prev_rmse = DBL_MAX, prev_fit = 0.0;                 /* :2115-2116 */
    for (int it = 0; it < cfg->max_iterations; ++it) {
        orc_pose6d_to_matrix(pose[0], pose[1], pose[2], pose[3], pose[4], pose[5], T);   /* :2164 */
        for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) R[i * 3 + j] = T[i * 4 + j]; t[i] = T[i * 4 + 3]; }
        prm.euler_rpy[0] = pose[0]; prm.euler_rpy[1] = pose[1]; prm.euler_rpy[2] = pose[2];
        orc_lin_out lo;
        orc_linearize(tree, src, n_src, stride, R, t, &prm, &lo, NULL);
        if (lo.n_eff < 10) { res->iterations = it; res->status = 1; break; }             /* :2272-2286 */
        const double fitness = (double)lo.n_pt / (double)n_src, rmse = sqrt(lo.sum_r2 / (double)lo.n_eff);   /* :2289-2291 */
        double H[36], x[6];
        orc_unpack_H(lo.H_upper, H);
        orc_analysis an;
        orc_analyze(H, detection, handling, cfg, &an);
        orc_solve(H, lo.g, handling, cfg, &an, x);
        int finite = 1;
        for (int i = 0; i < 6; ++i) if (!isfinite(x[i])) finite = 0;
        if (!finite) { res->iterations = it; res->status = 2; break; }                   /* :2605-2617 */
        for (int i = 0; i < 6; ++i) pose[i] += x[i];                                     /* :2633-2638 */
        const double d_rmse = rmse - prev_rmse, d_fit = fitness - prev_fit;              /* :2645-2648 */
        prev_rmse = rmse; prev_fit = fitness;
        if (log && it < log_cap) {
            orc_iter_log *L = &log[it];
            memset(L, 0, sizeof(*L));
            L->iter = it; L->n_eff = lo.n_eff; L->n_pt = lo.n_pt; L->fitness = fitness; L->rmse = rmse;
            L->objective = 0.5 * lo.sum_b2;
            for (int i = 0; i < 6; ++i) { L->gradient[i] = -lo.g[i]; L->dx[i] = x[i]; }
            orc_pose6d_to_matrix(pose[0], pose[1], pose[2], pose[3], pose[4], pose[5], L->T);   /* :2675 */
            orc_pose_error(cfg->gt, L->T, &L->trans_err, &L->rot_err_deg);
            memcpy(L->H_upper, lo.H_upper, sizeof(lo.H_upper));
            L->an = an;
        }
        res->iterations = it + 1;
        if (fabs(d_rmse) < 1e-4 && fabs(d_fit) < 1e-4) { res->converged = 1; break; }    /* :2679-2687 */
    }
    orc_pose6d_to_matrix(pose[0], pose[1], pose[2], pose[3], pose[4], pose[5], T);
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) res->R[i * 3 + j] = T[i * 4 + j]; res->t[i] = T[i * 4 + 3]; }
    if (final_pose6d) memcpy(final_pose6d, pose, sizeof(pose));
    return res->converged;      /* covariance (:2695-2738) is not restated here: host-only post-processing */
}

/* ============================================================================================
 * calculatePointToPointError (utils.hpp:538-589)
 * ==========================================================================================*/
void orc_p2p_error(const float *aligned, int64_t n_a, const orc_kdtree *ttree, const float *target,
                   int64_t n_t, double thr, double *rmse, double *fitness, double *chamfer, int64_t *valid) {
    double sum_sq = 0.0, sum_fwd = 0.0, sum_bwd = 0.0;
    int64_t nvalid = 0;
#ifdef _OPENMP
#pragma omp parallel for reduction(+ : sum_sq, sum_fwd, nvalid) schedule(static)
#endif
    for (int64_t i = 0; i < n_a; ++i) {
        int32_t id; float d2;
        if (orc_knn(ttree, aligned + 3 * i, 1, &id, &d2) > 0) {
            double dist = (double)sqrtf(d2);   /* std::sqrt(float) -> float, utils.hpp:557 */
            sum_fwd += dist;
            if (dist < thr) { sum_sq += (double)d2; nvalid++; }
        }
    }
    orc_kdtree *atree = orc_kdtree_build(aligned, n_a, 3);
#ifdef _OPENMP
#pragma omp parallel for reduction(+ : sum_bwd) schedule(static)
#endif
    for (int64_t i = 0; i < n_t; ++i) {
        int32_t id; float d2;
        if (orc_knn(atree, target + 3 * i, 1, &id, &d2) > 0) sum_bwd += (double)sqrtf(d2);
    }
    orc_kdtree_free(atree);
    *rmse = sqrt(sum_sq / (double)n_a);
    *fitness = (double)nvalid / (double)n_a;
    *chamfer = (sum_fwd / (double)n_a + sum_bwd / (double)n_t) / 2.0;
    *valid = nvalid;
}

size_t orc_sizeof_iter_log(void) { return sizeof(orc_iter_log); }
size_t orc_sizeof_analysis(void) { return sizeof(orc_analysis); }
