/*
 * dcreg_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A dependency-free C11/OpenMP restatement of the reference's point-to-plane ICP
 * hot path and its 6x6 degeneracy analysis / solvers.  It exists only so that
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can check /
 * time the HIP product path against it.  Nothing under dcreg_amd/ may include,
 * link or dlopen anything from oracle/.
 *
 * Parity status: PINNED.  The restatement reproduces the reference's own committed
 * per-iteration traces (tests/golden/release, tests/golden/paper, tests/golden/fig8;
 * see tests/test_oracle_golden.py).  The reference itself cannot be compiled here
 * (needs Eigen, PCL/FLANN, yaml-cpp, Ceres, Open3D, TBB: DCReg/CMakeLists.txt:11-21),
 * so there is no oracle/_ref build.
 *
 * Reference lines restated (paths relative to /root/reference/DCReg):
 *   src/icp_test_runner.cpp:1611-2060   Point2PlaneICP_SO3_OpenMP (engine loop)
 *   src/icp_test_runner.cpp:2418-2469   Schur complement block (Euler engine)
 *   include/dcreg.hpp:45-264            analyzeDegeneracy / solveDegenerateSystem
 *   include/math_utils.hpp:11-33,102-121,158-166   skew / exp / Jacobian / boxplus
 *   include/utils.hpp:452-535,538-589,630-636      pose helpers / p2p metrics / pointBodyToGlobal
 * Third-party arithmetic restated from its published algorithm (not vendored in the
 * reference): Eigen 3.3.7 ColPivHouseholderQR / makeHouseholder / FullPivLU rank
 * rule, FLANN exact k-NN with float L2 distances (PCL 1.10 KdTreeFLANN).
 */
#ifndef DCREG_ORACLE_H
#define DCREG_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* enum values follow the declaration order of include/utils.hpp:106-121 */
enum {
    ORC_DET_NONE = 0,
    ORC_DET_SCHUR_CONDITION_NUMBER = 1,
    ORC_DET_FULL_EVD_MIN_EIGENVALUE = 2,
    ORC_DET_EVD_SUB_CONDITION = 3,
    ORC_DET_FULL_SVD_CONDITION = 4
};
enum {
    ORC_HAND_NONE = 0,
    ORC_HAND_STANDARD_REGULARIZATION = 1,
    ORC_HAND_ADAPTIVE_REGULARIZATION = 2,
    ORC_HAND_PRECONDITIONED_CG = 3,
    ORC_HAND_SOLUTION_REMAPPING = 4,
    ORC_HAND_TRUNCATED_SVD = 5
};

typedef struct orc_kdtree orc_kdtree;

/* per-iteration linearisation parameters (icp_test_runner.cpp:1725,1750,1772,1776,1785,1691) */
typedef struct orc_lin_params {
    double search_radius;
    double max_plane_thickness_sq; /* 0.04 */
    double min_normal_norm;        /* 1e-6 */
    double weight_slope;           /* 0.9  */
    double weight_min;             /* 0.1  */
    int use_weight_derivative;     /* 0 = released source, 1 = paper traces */
    int num_threads;               /* 0 = OpenMP default; reference hard-codes 8 */
    int parameterization;          /* 0 = SO(3) right perturbation (math_utils.hpp:102-121), 1 = the Euler row of the second engine
                                      as the reference writes it (icp_test_runner.cpp:2299-2346: LOAM's brackets times coeff z,x,y),
                                      2 = the exact roll / pitch / yaw derivative (LOAM's own coefficient order; not in the reference) */
    int reserved_;
    double euler_rpy[3];           /* parameterization 1, 2: roll, pitch, yaw the R was built from (utils.hpp:452-460) */
} orc_lin_params;

typedef struct orc_lin_out {
    double H_upper[21]; /* row-major upper triangle, order [wx wy wz x y z] (hessian_computer.h:89-94) */
    double g[6];        /* A^T b ; the reference logs gradient = -g */
    double sum_r2;      /* sum r^2 over effective points (icp_test_runner.cpp:1803) */
    double sum_b2;      /* sum (f32(s r))^2 ; objective = 0.5*sum_b2 (icp_test_runner.cpp:1919) */
    int64_t n_eff;      /* correspondence_count    (icp_test_runner.cpp:1802) */
    int64_t n_pt;       /* correspondence_pt_count (icp_test_runner.cpp:1731) */
} orc_lin_out;

/* Optional per-point dump (any pointer may be NULL). flag: 1 valid, 0 radius/knn fail,
 * 2 |x|<min_normal_norm, 3 plane thickness, 4 weight<=weight_min. */
typedef struct orc_lin_debug {
    int32_t *nn_idx;  /* [5*n] target indices, ascending distance */
    float *nn_d2;     /* [5*n] */
    uint8_t *flag;    /* [n] */
    double *normal;   /* [3*n] unit normal */
    double *r;        /* [n] raw residual */
    double *s;        /* [n] weight */
} orc_lin_debug;

/* ICPParameters + Config subset (utils.hpp:82-171) */
typedef struct orc_config {
    double search_radius;
    int max_iterations;
    double thresh_rot, thresh_trans;       /* CONVERGENCE_THRESH_ROT / _TRANS */
    double thres_cond, thres_eig;          /* DEGENERACY_THRES_COND / _EIG */
    double kappa_target, pcg_tolerance;
    int pcg_max_iter;
    double std_reg_gamma;
    double adaptive_reg_alpha;
    int use_weight_derivative;
    int always_compute_schur;  /* 1: fill Schur/diag condition numbers for every method (paper traces) */
    int num_threads;
    int euler_exact_jacobian;  /* orc_icp_run_euler: 0 (default) = the reference's row (parameterization 1), 1 = parameterization 2 */
    double gt[16];             /* row-major 4x4 ground truth */
} orc_config;

typedef struct orc_analysis {
    int is_degenerate;
    int mask[6];
    double cond_schur_rot, cond_schur_trans;
    double cond_diag_rot, cond_diag_trans;
    double cond_full;
    double cond_full_sub_rot, cond_full_sub_trans;
    double eigenvalues_full[6];   /* ascending */
    double eigenvectors_full[36]; /* row-major, column i = eigenvector i */
    double singular_values[6];    /* descending */
    double lambda_schur_rot[3], lambda_schur_trans[3];
    double lambda_sub_rot[3], lambda_sub_trans[3];
    double schur_V_rot[9], schur_V_trans[9]; /* row-major, columns = eigenvectors */
    double P_preconditioner[36];
    int pcg_iterations;
} orc_analysis;

/* one row per ICP iteration (IterationLogData, utils.hpp:174-249) */
typedef struct orc_iter_log {
    int iter;
    int64_t n_eff, n_pt;
    double rmse, fitness, objective;
    double gradient[6]; /* = -A^T b */
    double dx[6];
    double T[16];       /* row-major, after the update */
    double trans_err, rot_err_deg;
    double H_upper[21];
    orc_analysis an;
} orc_iter_log;

typedef struct orc_icp_result {
    int converged;
    int iterations;
    int status;         /* 0 ok, 1 aborted: n_eff<10, 2 aborted: non-finite dx */
    double R[9], t[3];  /* final state */
    double cov[36];
} orc_icp_result;

/* ---- kd-tree (stands for pcl::KdTreeFLANN, utils.hpp:403) ---- */
orc_kdtree *orc_kdtree_build(const float *xyz, int64_t n, int64_t stride_floats);
void orc_kdtree_free(orc_kdtree *);
int64_t orc_kdtree_size(const orc_kdtree *);
/* exact k-NN, float squared distances, ascending; ties -> lower index. returns #found (<=k) */
int orc_knn(const orc_kdtree *, const float q[3], int k, int32_t *idx, float *d2);
void orc_knn_batch(const orc_kdtree *, const float *q, int64_t n, int64_t stride_floats, int k,
                   int32_t *idx, float *d2, int num_threads);

/* ---- small dense algebra (Eigen restatements) ---- */
/* x = argmin |A x - b| by column-pivoted Householder QR with Eigen's nonzeroPivots truncation.
 * A is m x n row-major, m<=8, n<=6. returns nonzero_pivots. */
int orc_colpiv_qr_solve(int m, int n, const double *A, const double *b, double *x);
/* symmetric eigen-decomposition (cyclic Jacobi), n<=6, ascending; V row-major, columns = vectors */
void orc_sym_eig(int n, const double *A, double *w, double *V);
/* 3x3 inverse by full-pivot LU with Eigen's isInvertible() rule; returns 1 if invertible */
int orc_inv3_fullpiv(const double *A, double *Ainv);
/* the same for n x n, n <= 6 (6x6: covariance, icp_test_runner.cpp:2016-2018) */
int orc_inv_fullpiv(int n, const double *A, double *Ainv);

/* ---- hot path ---- */
int orc_plane_fit(const double Q[15], double n_out[3], double *d_out, double *ps_out);
int orc_linearize(const orc_kdtree *, const float *src_xyz, int64_t n_src, int64_t stride_floats,
                  const double R[9], const double t[3], const orc_lin_params *,
                  orc_lin_out *, orc_lin_debug *dbg);

/* ---- 6x6 analysis / solve ---- */
void orc_unpack_H(const double H_upper[21], double H[36]);
void orc_analyze(const double H[36], int detection, int handling, const orc_config *, orc_analysis *);
void orc_solve(const double H[36], const double g[6], int handling, const orc_config *,
               orc_analysis *, double x[6]);
void orc_pcg(const double A[36], const double b[6], const double P[36], int max_iter, double tol,
             double x[6], int *iters);

/* ---- SE(3) / pose helpers ---- */
void orc_so3_exp(const double w[3], double R[9]);
void orc_boxplus(const double R[9], const double t[3], const double dx[6], double R2[9], double t2[3]);
void orc_pose6d_to_matrix(double roll, double pitch, double yaw, double x, double y, double z, double T[16]);
void orc_pose_error(const double gt[16], const double T[16], double *trans, double *rot_deg);

/* ---- engine ---- */
void orc_default_config(orc_config *);
int orc_icp_run(const orc_kdtree *, const float *src_xyz, int64_t n_src, int64_t stride_floats,
                const double R0[9], const double t0[3], int detection, int handling,
                const orc_config *, orc_iter_log *log, int log_capacity, orc_icp_result *);

/* calculatePointToPointError (utils.hpp:538-589) */
/* second engine, TestRunner::Point2PlaneICP (icp_test_runner.cpp:2064-2830): Pose6D state {roll,pitch,yaw,x,y,z}, LOAM
 * Jacobian, additive update, convergence on |d rmse|,|d fitness| < 1e-4.  The analysis / handling step is orc_analyze /
 * orc_solve (the reference inlines a copy).  NOT pinned by any committed trace of the reference. */
int orc_icp_run_euler(const orc_kdtree *, const float *src_xyz, int64_t n_src, int64_t stride_floats,
                      const double pose6d[6], int detection, int handling, const orc_config *,
                      orc_iter_log *log, int log_capacity, orc_icp_result *, double final_pose6d[6]);
/* dR/droll, dR/dpitch, dR/dyaw (row-major 3x3 each) of R = Rz(yaw) Ry(pitch) Rx(roll) */
void orc_euler_dR(double roll, double pitch, double yaw, double dR[27]);
/* the Euler row of one point p (body frame, float) and one float-stored weighted normal c: as the reference writes it
 * (icp_test_runner.cpp:2299-2346, parameterization 1) and the exact derivative (parameterization 2) */
void orc_euler_rows(const double rpy[3], const float p[3], const float c[3], double literal[6], double exact[6]);

void orc_p2p_error(const float *aligned_xyz, int64_t n_a, const orc_kdtree *target_tree,
                   const float *target_xyz, int64_t n_t, double error_threshold,
                   double *rmse, double *fitness, double *chamfer, int64_t *valid);

size_t orc_sizeof_iter_log(void);
size_t orc_sizeof_analysis(void);

#ifdef __cplusplus
}
#endif
#endif
