/*
 * dcreg.h -- C-ABI of the MI355X-native DCReg hot path (libdcreg_hip.so).
 *
 * The reference (JokerJohn/DCReg) has no FFI layer: the point-to-plane ICP inner loop sits behind two
 * C++ seams.  This header is the drop-in boundary a maintainer binds instead:
 *
 *   device seam  (steps 1-5 of one ICP iteration, DCReg/src/icp_test_runner.cpp:1704-1919)
 *       ICPContext::setTargetCloud           DCReg/include/utils.hpp:393-424     -> dcreg_set_target
 *       measure_cloud argument               DCReg/include/icp_test_runner.h:92  -> dcreg_set_source
 *       correspondence + plane fit + A,b + AtA/Atb   icp_test_runner.cpp:1714-1915 -> dcreg_linearize
 *   solver seam  (host, 6x6)
 *       DCReg::analyzeDegeneracy             DCReg/include/dcreg.hpp:45-166      -> dcreg_analyze_degeneracy
 *       DCReg::solveDegenerateSystem         DCReg/include/dcreg.hpp:168-264     -> dcreg_solve_degenerate_system
 *   engine seam
 *       TestRunner::Point2PlaneICP_SO3_OpenMP  icp_test_runner.h:92-102, icp_test_runner.cpp:1611-2060
 *                                                                                 -> dcreg_icp_run
 *       TestRunner::Point2PlaneICP (Euler / LOAM parameterisation)  icp_test_runner.h:72-82, icp_test_runner.cpp:2064-2830
 *                                                                                 -> dcreg_icp_run_euler
 *       TestRunner::runMethod num_runs loop  icp_test_runner.cpp:331-390         -> dcreg_icp_run_trials
 *       calculatePointToPointError           utils.hpp:538-589                   -> dcreg_p2p_error
 *
 * Conventions: plain pointers and sizes only; the caller owns host buffers (borrowed for the call);
 * a ctx owns its device memory, stream and events; return 0 = ok, <0 = error; a ctx is
 * single-threaded (one per GPU / stream), several may run concurrently.  Results are deterministic
 * (fixed reduction order, no floating-point atomics).  There is NO CPU fallback: without a usable
 * HIP device every device-seam call fails with DCREG_E_DEVICE.
 */
#ifndef DCREG_H
#define DCREG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DCREG_OK 0
#define DCREG_E_INVALID (-1)
#define DCREG_E_NOMEM (-2)
#define DCREG_E_DEVICE (-3)
#define DCREG_E_STATE (-4)

/* DetectionMethod / HandlingMethod: numeric values follow DCReg/include/utils.hpp:106-121 */
enum dcreg_detection {
    DCREG_NONE_DETE = 0,
    DCREG_SCHUR_CONDITION_NUMBER = 1,
    DCREG_FULL_EVD_MIN_EIGENVALUE = 2,
    DCREG_EVD_SUB_CONDITION = 3,
    DCREG_FULL_SVD_CONDITION = 4
};
enum dcreg_handling {
    DCREG_NONE_HAND = 0,
    DCREG_STANDARD_REGULARIZATION = 1,
    DCREG_ADAPTIVE_REGULARIZATION = 2,
    DCREG_PRECONDITIONED_CG = 3,
    DCREG_SOLUTION_REMAPPING = 4,
    DCREG_TRUNCATED_SVD = 5
};

typedef struct dcreg_ctx dcreg_ctx;

/* constants of one linearisation; defaults = the literals in icp_test_runner.cpp */
typedef struct dcreg_lin_params {
    double search_radius;          /* Config::search_radius, icp_test_runner.cpp:1725 */
    double max_plane_thickness_sq; /* 0.2*0.2, :1772 */
    double min_normal_norm;        /* 1e-6,    :1750 */
    double weight_slope;           /* 0.9,     :1776 */
    double weight_min;             /* 0.1,     :1785 */
    int use_weight_derivative;     /* USE_WEIGHT_DERIVATIVE, :1691 (0 = released source, 1 = paper) */
    int k;                         /* 5 (only value supported) */
    int parameterization;          /* enum dcreg_parameterization below */
    int reserved_;
    double euler_rpy[3];           /* DCREG_PARAM_EULER / _EULER_EXACT: roll, pitch, yaw of the pose the R passed alongside was built
                                      from (Pose6D2Matrix: R = Rz(yaw) Ry(pitch) Rx(roll), utils.hpp:452-460) */
} dcreg_lin_params;

/* DCREG_PARAM_SO3         right perturbation on SO(3), math_utils.hpp:102-121 (first engine, icp_test_runner.cpp:1863-1907);
 * DCREG_PARAM_EULER       the roll / pitch / yaw row of the second engine AS THE REFERENCE WRITES IT, icp_test_runner.cpp:2299-2346:
 *                         LOAM's three brackets per angle, multiplied by coeff.z, coeff.x, coeff.y (:2323-2335) where LOAM / LIO-SAM
 *                         multiply by coeff.x, coeff.y, coeff.z.  With the reference's order the rotation columns are not the
 *                         derivative of the residual; this value reproduces the reference, term by term;
 * DCREG_PARAM_EULER_EXACT additive, not in the reference: the exact derivative of c . (Rz(yaw) Ry(pitch) Rx(roll) p) by roll /
 *                         pitch / yaw (= LOAM's coefficient order). */
enum dcreg_parameterization { DCREG_PARAM_SO3 = 0, DCREG_PARAM_EULER = 1, DCREG_PARAM_EULER_EXACT = 2 };

typedef struct dcreg_lin_out {
    double H_upper[21]; /* A^T A, row-major upper triangle, order [wx wy wz x y z] (hessian_computer.h:89-94) */
    double g[6];        /* A^T b  (the reference logs gradient = -g, :1918) */
    double sum_r2;      /* sum r^2 over effective points (:1803) -> rmse */
    double sum_b2;      /* sum (float(s r))^2 -> objective = 0.5*sum_b2 (:1919) */
    int64_t n_eff;      /* correspondence_count (:1802) */
    int64_t n_pt;       /* correspondence_pt_count (:1731) -> fitness */
} dcreg_lin_out;

typedef struct dcreg_index_info {
    double cell;        /* grid cell edge (m) */
    double origin[3];
    int32_t dims[3];
    int64_t n_cells;
    int64_t n_target;
    int64_t n_source;
    int32_t max_ring;   /* rings needed to cover search_radius of the last linearisation */
} dcreg_index_info;

/* ---------------- device seam ---------------- */
int dcreg_backend_create(dcreg_ctx **out, int device);
void dcreg_backend_destroy(dcreg_ctx *);
const char *dcreg_last_error(const dcreg_ctx *);
/* use an existing hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL = ctx-owned stream */
int dcreg_set_stream(dcreg_ctx *, void *hip_stream);
/* options (every one of them changes speed only: results are identical whatever their values)
 *   "warm_start"    1 (default) = keep, per source point, the neighbours its last search found and a certificate of how far the
 *                   point may move before the nearest five can change; later linearisations skip the search of every point whose
 *                   certificate still holds and bound the searches that remain by the old neighbours.  0 = search every point
 *                   from scratch in every call;
 *   "cert_margin"   default 0.05: searches cover search_radius * (1 + margin), so that "the 5th neighbour is beyond the radius" can
 *                   be certified too (takes effect at the next dcreg_set_target: the grid cells follow the search radius);
 *   "cert_inflate"  default 0.005: searches among nearby points look 0.5 % further than they must, which yields the second kind of
 *                   certificate ("the five nearest are among these six"; search.hpp);
 *   "fast_plane_fit" 1 (default) = the reduced-instruction 5x3 plane fit; 0 = the Eigen-shaped factorisation step for step (planes
 *                   agree to a few ulp, gate flags are identical on every test scene);
 *   "spin"          1 (default) = wait for results on the pinned result flags instead of hipStreamSynchronize;
 *   "wait_seconds"  default 30: how long a result is awaited before the stream is drained to look for a device fault;
 *   "dispatch_order" 1 (default) = launches with more query blocks than the device holds at once hand them out heaviest group
 *                   first while dcreg_hint_misalignment says the clouds are misaligned (kernels.hpp k_group_cost); 0 = index order;
 *   "far_bound"     1 (default) = a query whose bound is loose (nothing known yet, or neighbours of a pose far away) starts its
 *                   search from the points around the nearest occupied cell (next dcreg_set_target);
 *   "cell", "cell_factor", "x_subdiv", "gap_field": the grid index (cell edge in metres, 0 = auto = cell_factor x the estimated
 *                   5th-neighbour distance; x sub-cells per cell 1..16, default 8; 1 = build the empty-space distance field) at the
 *                   next dcreg_set_target;
 *   "max_table_entries" default 2^30 (at most 2^31): entries of the dense cell table - one uint32 per x sub-cell of the target's
 *                   bounding box; a map whose cells at the wanted edge would need more gets fewer x sub-cells first and a larger cell
 *                   edge after that (next dcreg_set_target);
 *   "roi_index", "roi_margin": the WINDOW index of a large map.  A map whose table ran into that budget is searched through cells that
 *                   grow with its extent; dcreg_linearize / the engines' single-pose launches therefore search such a map through a
 *                   second index over the map's points inside a box around the transformed source cloud (its bounding box at the pose +
 *                   the search radius + roi_margin metres, default 20) - the same neighbours and bitwise the same sums, cells sized for
 *                   the local density.  Built by the first linearisation (about 4 ms, whatever the map's size), kept until a
 *                   pose leaves the box, then rebuilt around that pose (a queued gated launch is called off: dcreg_linearize_gate_open
 *                   returns DCREG_E_STATE and the caller starts the launch with dcreg_linearize_batch_begin, as the engines do).
 *                   roi_index 1 (default) = for maps whose cell edge the budget enlarged by more than one step (x 1.26) or whose x sub-cells it took, 0 = never, 2 = always.  dcreg_knn,
 *                   dcreg_p2p_error, batched launches and debug dumps always run on the whole map; dcreg_index_info_get describes
 *                   the whole map's index (dcreg_debug.h dcreg_roi_info: the window).
 * Profiling / experiment knobs are listed in dcreg_debug.h. */
int dcreg_set_option(dcreg_ctx *, const char *key, double value);
/* target cloud: copies + builds the device spatial index (stands for kd-tree build, utils.hpp:403).
 * search_radius_hint bounds the cell size (cell <= radius); pass Config::search_radius. */
int dcreg_set_target(dcreg_ctx *, const float *xyz, int64_t n, int64_t stride_floats, double search_radius_hint);
int dcreg_set_target_device(dcreg_ctx *, const float *d_xyz, int64_t n, int64_t stride_floats, double search_radius_hint);
/* source ("measure") cloud: copies, orders along a space-filling curve.  The caller's buffer is consumed when the call returns - it
 * may be reused or freed at once, whatever kind of host memory it is.  A frame of at most 65536 points (and 2^20 floats) is copied
 * into the context's own pinned block and queued from there without a stream synchronise (the registration path: the first
 * linearisation runs behind the sort): a device fault of the upload or the sort then surfaces at that linearisation, not here.
 * Clouds with non-finite coordinates are refused (DCREG_E_INVALID), source and target alike. */
int dcreg_set_source(dcreg_ctx *, const float *xyz, int64_t n, int64_t stride_floats);
int dcreg_set_source_device(dcreg_ctx *, const float *d_xyz, int64_t n, int64_t stride_floats);
int dcreg_default_lin_params(dcreg_lin_params *, double search_radius);
/* one ICP linearisation (steps 1-5): R row-major 3x3, t 3 */
int dcreg_linearize(dcreg_ctx *, const double R[9], const double t[3], const dcreg_lin_params *, dcreg_lin_out *);
/* the same for n_poses independent poses of the same cloud pair in ONE launch (Monte-Carlo trials) */
int dcreg_linearize_batch(dcreg_ctx *, int n_poses, const double *R9, const double *t3, const dcreg_lin_params *,
                          dcreg_lin_out *outs);
/* asynchronous pair of the above: _begin queues the copy + kernels on the ctx's stream and returns, _end waits for the
 * results of that slot (pinned-memory sequence numbers) and unpacks them.  Two slots (0, 1) with their own buffers: keep
 * one batch on the device while the host solves the other (dcreg_icp_run_trials does).  R9 / t3 are copied by _begin. */
int dcreg_linearize_batch_begin(dcreg_ctx *, int slot, int n_poses, const double *R9, const double *t3, const dcreg_lin_params *);
/* pipelined single-pose launches (what dcreg_icp_run does between two iterations): _gated_begin queues a linearisation on `slot`
 * whose pose is not known yet - a one-wave gate kernel in front of it waits for it - typically while the previous linearisation
 * still runs; _gate_open publishes the pose (two stores, the device starts at once: no launch on the critical path); _gate_abort
 * calls the queued linearisation off (it returns without touching results or warm state).  Exactly one of the two must follow every
 * _gated_begin, before anything else is queued on the context (other launches are refused meanwhile); results come through
 * dcreg_linearize_batch_end(slot).  Needs the default "spin" option (a stream synchronise would wait for the gate).  A gate nobody
 * opens gives up after two minutes. */
int dcreg_linearize_gated_begin(dcreg_ctx *, int slot, const dcreg_lin_params *);
int dcreg_linearize_gate_open(dcreg_ctx *, const double R[9], const double t[3]);
int dcreg_linearize_gate_abort(dcreg_ctx *);
int dcreg_linearize_batch_end(dcreg_ctx *, int slot, dcreg_lin_out *outs);
/* Neighbour states for batched launches.  A single-pose linearisation reuses what its own previous call found (neighbours +
 * certificates, kept inside the ctx); poses of a batch belong to different trajectories, so each needs a state of its own:
 * reserve n_states of them (76 B per source point each; nothing is cleared - a state counts as empty until its first launch
 * has filled it), then name the state of every pose in state_ids (0 <= id < n_states, each id at most once per launch,
 * -1 = search from scratch, keep nothing).  A state is read and updated by the launch, so consecutive launches of one
 * Monte-Carlo trial under the same id skip the searches their certificates cover.  dcreg_reset_warm_state marks one state
 * empty again (a trial slot that takes the next trial); state_id = -1 names the context's OWN state, the one single-pose launches and
 * dcreg_icp_run keep: after it the next launch searches every point from scratch, as the first launch after dcreg_set_source does
 * (the reference builds a fresh ICPContext for every run, icp_test_runner.cpp:408-409: bench.py's `cold_run`).  Results are identical
 * with or without states.
 * dcreg_set_target / dcreg_set_source drop all states. */
/* Scheduling hint, never needed for correctness: how far (metres, roughly) the source points are expected to lie from the map at the
 * poses of the next single-pose linearisations - e.g. the RMS residual of the last iteration.  While it is above half a grid cell
 * (and the context's own cost estimate of the cloud pair is uneven) the query blocks of a launch are handed out heaviest group first
 * instead of in index order (longest processing time first: a launch ends when its slowest block does); while it is above a cell and
 * a half, launches of large clouds most of whose points search run the linearisation kernel in one-wave blocks (kernels.hpp
 * k_lin<.., ONE>: worth a tenth of such a launch, a few microseconds lost on others).  The 31 sums do not depend on it.  Default: +infinity (no knowledge: assume misaligned).  dcreg_icp_run* call it themselves. */
int dcreg_hint_misalignment(dcreg_ctx *, double metres);
int dcreg_reserve_warm_states(dcreg_ctx *, int64_t n_states);
int dcreg_reset_warm_state(dcreg_ctx *, int64_t state_id);
int dcreg_linearize_batch_begin_warm(dcreg_ctx *, int slot, int n_poses, const double *R9, const double *t3,
                                     const int32_t *state_ids, const dcreg_lin_params *);
/* exact k-NN (k = 1 or 5) of host queries against the target index; float sq. distances, (d2, idx) order */
int dcreg_knn(dcreg_ctx *, const float *q_xyz, int64_t n, int64_t stride_floats, int k, double max_radius,
              int32_t *idx, float *d2);
int dcreg_index_info_get(const dcreg_ctx *, dcreg_index_info *);

/* ---------------- solver seam (host only, no device needed) ---------------- */
/* Config + ICPParameters subset (utils.hpp:82-171) */
typedef struct dcreg_config {
    double search_radius;
    int max_iterations;
    double CONVERGENCE_THRESH_ROT, CONVERGENCE_THRESH_TRANS;
    double DEGENERACY_THRES_COND, DEGENERACY_THRES_EIG;
    double KAPPA_TARGET, PCG_TOLERANCE;
    int PCG_MAX_ITER;
    double STD_REG_GAMMA, ADAPTIVE_REG_ALPHA;
    int use_weight_derivative;  /* additive key: icp_test_runner.cpp:1691 as a switch */
    int always_compute_schur;   /* additive key: fill Schur/diag numbers for every method (paper traces) */
    int euler_exact_jacobian;   /* additive key, dcreg_icp_run_euler only: 0 (default) = the reference's row (DCREG_PARAM_EULER),
                                   1 = the exact derivative (DCREG_PARAM_EULER_EXACT) */
    int reserved_cfg_;
    double gt_matrix[16];       /* row-major */
} dcreg_config;

/* DegeneracyAnalysisResult (utils.hpp:427-448) */
typedef struct dcreg_analysis {
    int isDegenerate;
    int degenerate_mask[6];
    double cond_schur_rot, cond_schur_trans;
    double cond_diag_rot, cond_diag_trans;
    double cond_full;
    double cond_full_sub_rot, cond_full_sub_trans;
    double eigenvalues_full[6];   /* ascending */
    double eigenvectors_full[36]; /* row-major, column i <-> eigenvalue i */
    double singular_values[6];    /* descending */
    double lambda_schur_rot[3], lambda_schur_trans[3];
    double lambda_sub_rot[3], lambda_sub_trans[3];
    double schur_V_rot[9], schur_V_trans[9];
    double aligned_V_rot[9], aligned_V_trans[9];
    int rot_indices[3], trans_indices[3];
    double P_preconditioner[36];
    double W_adaptive[36];
    int pcg_iterations;
} dcreg_analysis;

void dcreg_default_config(dcreg_config *);
int dcreg_analyze_degeneracy(const double H[36], int detection, int handling, const dcreg_config *, dcreg_analysis *);
int dcreg_solve_degenerate_system(const double H[36], const double g[6], int handling, const dcreg_config *,
                                  dcreg_analysis *, double x[6]);
void dcreg_unpack_hessian(const double H_upper[21], double H[36]);
void dcreg_boxplus(const double R[9], const double t[3], const double dx[6], double R_out[9], double t_out[3]);
void dcreg_pose6d_to_matrix(double roll, double pitch, double yaw, double x, double y, double z, double T[16]);
void dcreg_pose_error(const double gt[16], const double T[16], double *trans_m, double *rot_deg);

/* ---------------- engine seam ---------------- */
/* IterationLogData (utils.hpp:174-249) */
typedef struct dcreg_iter_log {
    int iter_count;
    int64_t effective_points, corr_pt_count;
    double rmse, fitness, objective_value;
    double gradient[6];
    double update_dx[6];
    double transform_matrix[16];
    double trans_error_vs_gt, rot_error_vs_gt;
    double iter_time_ms;
    double H_upper[21];
    dcreg_analysis analysis;
} dcreg_iter_log;

typedef struct dcreg_icp_result {
    int converged;
    int iterations;
    int status;        /* 0 ok, 1 n_eff<10 abort (:1847), 2 non-finite dx abort (:1942), 3 bad input (:1635) */
    double R[9], t[3];
    double icp_cov[36];
    double time_ms;
} dcreg_icp_result;

int dcreg_icp_run(dcreg_ctx *, const double R0[9], const double t0[3], int detection, int handling,
                  const dcreg_config *, dcreg_iter_log *log, int log_capacity, dcreg_icp_result *);

/* n independent scan pairs at once: one host thread per ctx (each ctx owns its clouds, index, stream), every thread runs
 * dcreg_icp_run.  One 100 k-point linearisation fills well under half of an MI355X and the device idles during each host
 * step, so independent pairs interleave: 4 pairs in flight give ~3.3x the one-pair iteration rate.  R0 = n x 9,
 * t0 = n x 3; results[i].status etc. as for dcreg_icp_run; returns the first non-OK code of any pair (all pairs still run
 * to completion).  Logs are not collected in this mode. */
int dcreg_icp_run_many(int n, dcreg_ctx *const *ctxs, const double *R0, const double *t0, int detection, int handling,
                       const dcreg_config *, dcreg_icp_result *results);

/* Point sharding of ONE scan pair over several devices: the ctx holds the whole target and THIS rank's slice
 * of the source; after every linearisation `reduce` must replace row[32] (21 H, 6 g, sum r^2, sum b^2, n_eff, n_pt, pad) by
 * the sum over all ranks, added in rank order so that every rank obtains bitwise the same totals (e.g. an all_gather over
 * RCCL + ordered sum; return 0 on success).  Every rank then takes the identical host step: no broadcast is needed.
 * n_source_total = points of the whole source cloud (fitness, :1856).  reduce == NULL: plain dcreg_icp_run. */
typedef int (*dcreg_reduce_fn)(double row[32], void *user);
int dcreg_icp_run_sharded(dcreg_ctx *, const double R0[9], const double t0[3], int detection, int handling,
                          const dcreg_config *, int64_t n_source_total, dcreg_reduce_fn reduce, void *reduce_user,
                          dcreg_iter_log *log, int log_capacity, dcreg_icp_result *);

/* The same with the exchange done natively: ONE ncclAllGather (RCCL over xGMI) of the 32-double rows per iteration on the
 * ctx's stream, rows added in rank order, inside the C++ engine loop (no callback).  Set-up: rank 0 obtains 128 opaque bytes
 * from dcreg_comm_unique_id and hands them to every rank by any means (the Python launcher broadcasts them with
 * torch.distributed); every rank then calls dcreg_comm_init(ctx, id, rank, world) - collectively, like ncclCommInitRank - and
 * dcreg_icp_run_sharded_rccl.  dcreg_comm_allgather_sum is the exchange step on its own.  RCCL is dlopen'ed on first use. */
int dcreg_comm_unique_id(void *id128);
int dcreg_comm_init(dcreg_ctx *, const void *id128, int rank, int world);
int dcreg_comm_destroy(dcreg_ctx *);
int dcreg_comm_allgather_sum(dcreg_ctx *, double row[32]);
int dcreg_icp_run_sharded_rccl(dcreg_ctx *, const double R0[9], const double t0[3], int detection, int handling,
                               const dcreg_config *, int64_t n_source_total, dcreg_iter_log *log, int log_capacity,
                               dcreg_icp_result *);

/* The second engine of the reference (selected by Config::use_so3_parameterization == false, icp_test_runner.cpp:443-458):
 * state = Pose6D {roll, pitch, yaw, x, y, z}, the Jacobian of :2299-2346 with the float-stored weighted normal and no weight
 * derivative - literally, coefficient permutation included (enum dcreg_parameterization above; dcreg_config::euler_exact_jacobian
 * selects the exact derivative instead) -, additive update (:2633-2638), convergence on |d rmse| < 1e-4 && |d fitness| < 1e-4
 * (:2679-2687), covariance mapped through the Euler->Lie Jacobian (:2695-2738).  pose6d = {roll, pitch, yaw, x, y, z};
 * final_pose6d receives the optimised pose (may be NULL).  The 6x6 analysis / handling step goes through the solver
 * seam above (the reference inlines a copy of it in this engine).  No committed trace of the reference exercises this
 * engine: parity is pinned on the shared correspondence steps only. */
int dcreg_icp_run_euler(dcreg_ctx *, const double pose6d[6], int detection, int handling, const dcreg_config *,
                        dcreg_iter_log *log, int log_capacity, dcreg_icp_result *, double final_pose6d[6]);

/* TestResult subset per trial (utils.hpp:253-303) */
typedef struct dcreg_trial_result {
    int converged, iterations, status;
    double time_ms;
    double trans_error_m, rot_error_deg;
    double final_rmse, final_fitness;
    int64_t corr_num;
    double final_transform[16];
    double H_upper[21];
    int degenerate_mask[6];
} dcreg_trial_result;

/* n_trials independent ICP runs of the same cloud pair from different initial poses (the num_runs loop of runMethod) as a
 * continuously refilled batch: up to 256 trials are in flight, every iteration of a group of them is ONE batched launch, and
 * a trial that ends hands its slot to the next one in line at once.  Each trial is bitwise the single run (dcreg_icp_run) of
 * its pose. */
int dcreg_icp_run_trials(dcreg_ctx *, int n_trials, const double *R0_9, const double *t0_3, int detection,
                         int handling, const dcreg_config *, dcreg_trial_result *results);

/* Initial pose of Monte-Carlo trial k (the reference has no RNG: its num_runs loop, icp_test_runner.cpp:339-349, repeats one
 * deterministic run; the seeded perturbation is this build's definition, shared by the runner and dcreg_amd/montecarlo.py):
 * k == 0 -> the base pose; k >= 1 -> base + U(-amp, amp) per degree of freedom from MT19937(low 32 bits of seed + k),
 * 53-bit doubles, drawn in the order x y z roll pitch yaw.  base = {x, y, z [m], roll, pitch, yaw [rad]}; T row-major 4x4
 * = Pose6D2Matrix (utils.hpp:452-460); pose_xyzrpy (may be NULL) receives the perturbed six numbers. */
int dcreg_trial_pose(const double base_xyzrpy[6], uint64_t seed, int64_t k, double trans_amp, double rot_amp_rad, double T[16],
                     double pose_xyzrpy[6]);

/* The Monte-Carlo experiment of one rank: trials k = first_trial + j * trial_stride, j = 0 .. n_trials - 1 (rank r of w: first_trial
 * = r, stride = w), initial poses from dcreg_trial_pose, run as dcreg_icp_run_trials does with `slots` trials in flight (0 = 256).
 * results[j] belongs to trial first_trial + j * trial_stride. */
int dcreg_icp_run_montecarlo(dcreg_ctx *, const double base_xyzrpy[6], uint64_t seed, int64_t first_trial, int64_t trial_stride,
                             int64_t n_trials, double trans_amp, double rot_amp_rad, int detection, int handling,
                             const dcreg_config *, int slots, dcreg_trial_result *results);

/* The Monte-Carlo experiment (BASELINE configs[4]; runMethod's num_runs loop + updateStatistics / finalizeStatistics,
 * icp_test_runner.cpp:331-390, 604-664) as ONE job over the ranks of the ctx's communicator (dcreg_comm_init; without one: a job of one
 * rank): this rank runs trials k = rank, rank + world, ... (dcreg_icp_run_montecarlo), the fixed-size trial records of all ranks are
 * gathered with ONE ncclAllGather (RCCL over xGMI) on the ctx's stream, and EVERY rank receives all n_trials records ordered by trial
 * and the method's statistics - no host-side collective, no Python.  A record is DCREG_TRIAL_RECORD_DOUBLES doubles:
 *   [0] converged [1] iterations [2] time_ms [3] trans_error_m [4] rot_error_deg [5] final_rmse [6] final_fitness [7] corr_num
 *   [8] status [9] trial index [10..25] final transform, row-major [26..46] last Hessian, upper triangle [47..52] degenerate mask.
 * records: [n_trials * DCREG_TRIAL_RECORD_DOUBLES] or NULL; stats may be NULL. */
#define DCREG_TRIAL_RECORD_DOUBLES 64
typedef struct dcreg_method_stats {        /* MethodStatistics, utils.hpp:305-330 */
    int64_t total_runs, converged_runs;
    double success_rate;
    double mean_trans_error, std_trans_error, min_trans_error, max_trans_error;     /* population std (:660-662) */
    double mean_rot_error, std_rot_error, min_rot_error, max_rot_error;
    double mean_time_ms, std_time_ms;
    double mean_iterations, mean_rmse, mean_fitness;
    int64_t corr_num;              /* correspondences of all final iterations */
    int64_t iterations_total;      /* ICP iterations of all trials */
    int ranks_seen;                /* ranks that contributed at least one record (= the communicator's size when every rank did) */
    int world;
} dcreg_method_stats;
int dcreg_montecarlo_job(dcreg_ctx *, const double base_xyzrpy[6], uint64_t seed, int64_t n_trials, double trans_amp, double rot_amp_rad,
                         int detection, int handling, const dcreg_config *, int slots, double *records, dcreg_method_stats *stats);
/* the gather on its own: count doubles of this rank -> recv[world * count], rank-major, on every rank; rank / size of the communicator */
int dcreg_comm_allgather(dcreg_ctx *, const double *send, double *recv, int64_t count);
int dcreg_comm_info(const dcreg_ctx *, int *rank, int *world);

/* host threads the batched engines may use for the per-trial 6x6 steps (OpenMP; the reference hard-codes 8, :1714).  Launchers
 * that pin OMP_NUM_THREADS=1 (torch.distributed.run) should set this to the CPUs the rank really owns.  Whatever is set, the engines
 * never use more threads than the process can keep busy - min(affinity mask, cgroup CPU quota): OpenMP's default inside a container is
 * the machine's hardware thread count. */
int dcreg_set_host_threads(int n);
int dcreg_get_host_threads(void);

/* calculatePointToPointError (utils.hpp:538-589): aligned = T * source (float), both directions on the GPU */
int dcreg_p2p_error(dcreg_ctx *, const double T[16], double error_threshold, double *rmse, double *fitness,
                    double *chamfer, int64_t *valid_correspondences);

size_t dcreg_sizeof(const char *struct_name);
const char *dcreg_version(void);

#ifdef __cplusplus
}
#endif
#endif /* DCREG_H */
