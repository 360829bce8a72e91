"""ctypes mirror of include/dcreg.h (the C-ABI of libdcreg_hip.so).

Python is plumbing here: tests, bench.py and multi-GPU launch use this thin binding; the product is the
shared library.  There is no CPU fallback: creating a Context without a usable HIP device raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DCREG_LIB") or os.path.join(_HERE, "lib", "libdcreg_hip.so")   # DCREG_LIB: an experiment variant (build.py)

# enum values of DCReg/include/utils.hpp:106-121
DETECTION = {"NONE_DETE": 0, "SCHUR_CONDITION_NUMBER": 1, "FULL_EVD_MIN_EIGENVALUE": 2,
             "EVD_SUB_CONDITION": 3, "FULL_SVD_CONDITION": 4}
HANDLING = {"NONE_HAND": 0, "STANDARD_REGULARIZATION": 1, "ADAPTIVE_REGULARIZATION": 2,
            "PRECONDITIONED_CG": 3, "SOLUTION_REMAPPING": 4, "TRUNCATED_SVD": 5}
# method-name dispatch of the YAML test_methods section (DCReg/config/icp.yaml:101-116)
METHODS = {
    "ME-SR": ("FULL_EVD_MIN_EIGENVALUE", "SOLUTION_REMAPPING"),
    "ME-TSVD": ("FULL_EVD_MIN_EIGENVALUE", "TRUNCATED_SVD"),
    "ME-TReg": ("FULL_EVD_MIN_EIGENVALUE", "STANDARD_REGULARIZATION"),
    "FCN-SR": ("FULL_SVD_CONDITION", "SOLUTION_REMAPPING"),
    "Ours": ("SCHUR_CONDITION_NUMBER", "PRECONDITIONED_CG"),
    "NONE": ("NONE_DETE", "NONE_HAND"),
}

OK, E_INVALID, E_NOMEM, E_DEVICE, E_STATE = 0, -1, -2, -3, -4


class LinParams(C.Structure):
    _fields_ = [("search_radius", C.c_double), ("max_plane_thickness_sq", C.c_double),
                ("min_normal_norm", C.c_double), ("weight_slope", C.c_double), ("weight_min", C.c_double),
                ("use_weight_derivative", C.c_int), ("k", C.c_int), ("parameterization", C.c_int), ("reserved_", C.c_int),
                ("euler_rpy", C.c_double * 3)]


class LinOut(C.Structure):
    _fields_ = [("H_upper", C.c_double * 21), ("g", C.c_double * 6), ("sum_r2", C.c_double),
                ("sum_b2", C.c_double), ("n_eff", C.c_int64), ("n_pt", C.c_int64)]


class LinDebug(C.Structure):
    _fields_ = [("nn_idx", C.POINTER(C.c_int32)), ("nn_d2", C.POINTER(C.c_float)),
                ("flag", C.POINTER(C.c_uint8)), ("normal", C.POINTER(C.c_double)),
                ("r", C.POINTER(C.c_double)), ("s", C.POINTER(C.c_double)), ("stats", C.POINTER(C.c_uint32)),
                ("stamps", C.POINTER(C.c_uint64))]


class LaunchStats(C.Structure):
    _fields_ = [("launches", C.c_int64), ("poses", C.c_int64), ("points", C.c_int64), ("points_searched", C.c_int64),
                ("points_team", C.c_int64)]


class MethodStats(C.Structure):
    _fields_ = [("total_runs", C.c_int64), ("converged_runs", C.c_int64), ("success_rate", C.c_double),
                ("mean_trans_error", C.c_double), ("std_trans_error", C.c_double), ("min_trans_error", C.c_double), ("max_trans_error", C.c_double),
                ("mean_rot_error", C.c_double), ("std_rot_error", C.c_double), ("min_rot_error", C.c_double), ("max_rot_error", C.c_double),
                ("mean_time_ms", C.c_double), ("std_time_ms", C.c_double),
                ("mean_iterations", C.c_double), ("mean_rmse", C.c_double), ("mean_fitness", C.c_double),
                ("corr_num", C.c_int64), ("iterations_total", C.c_int64), ("ranks_seen", C.c_int), ("world", C.c_int)]


class IndexInfo(C.Structure):
    _fields_ = [("cell", C.c_double), ("origin", C.c_double * 3), ("dims", C.c_int32 * 3),
                ("n_cells", C.c_int64), ("n_target", C.c_int64), ("n_source", C.c_int64),
                ("max_ring", C.c_int32)]


class Config(C.Structure):
    _fields_ = [("search_radius", C.c_double), ("max_iterations", C.c_int),
                ("CONVERGENCE_THRESH_ROT", C.c_double), ("CONVERGENCE_THRESH_TRANS", C.c_double),
                ("DEGENERACY_THRES_COND", C.c_double), ("DEGENERACY_THRES_EIG", C.c_double),
                ("KAPPA_TARGET", C.c_double), ("PCG_TOLERANCE", C.c_double), ("PCG_MAX_ITER", C.c_int),
                ("STD_REG_GAMMA", C.c_double), ("ADAPTIVE_REG_ALPHA", C.c_double),
                ("use_weight_derivative", C.c_int), ("always_compute_schur", C.c_int),
                ("euler_exact_jacobian", C.c_int), ("reserved_cfg_", C.c_int),
                ("gt_matrix", C.c_double * 16)]


class Analysis(C.Structure):
    _fields_ = [("isDegenerate", C.c_int), ("degenerate_mask", C.c_int * 6),
                ("cond_schur_rot", C.c_double), ("cond_schur_trans", C.c_double),
                ("cond_diag_rot", C.c_double), ("cond_diag_trans", C.c_double),
                ("cond_full", C.c_double), ("cond_full_sub_rot", C.c_double),
                ("cond_full_sub_trans", C.c_double), ("eigenvalues_full", C.c_double * 6),
                ("eigenvectors_full", C.c_double * 36), ("singular_values", C.c_double * 6),
                ("lambda_schur_rot", C.c_double * 3), ("lambda_schur_trans", C.c_double * 3),
                ("lambda_sub_rot", C.c_double * 3), ("lambda_sub_trans", C.c_double * 3),
                ("schur_V_rot", C.c_double * 9), ("schur_V_trans", C.c_double * 9),
                ("aligned_V_rot", C.c_double * 9), ("aligned_V_trans", C.c_double * 9),
                ("rot_indices", C.c_int * 3), ("trans_indices", C.c_int * 3),
                ("P_preconditioner", C.c_double * 36), ("W_adaptive", C.c_double * 36),
                ("pcg_iterations", C.c_int)]


class IterLog(C.Structure):
    _fields_ = [("iter_count", C.c_int), ("effective_points", C.c_int64), ("corr_pt_count", C.c_int64),
                ("rmse", C.c_double), ("fitness", C.c_double), ("objective_value", C.c_double),
                ("gradient", C.c_double * 6), ("update_dx", C.c_double * 6),
                ("transform_matrix", C.c_double * 16), ("trans_error_vs_gt", C.c_double),
                ("rot_error_vs_gt", C.c_double), ("iter_time_ms", C.c_double),
                ("H_upper", C.c_double * 21), ("analysis", Analysis)]


class IcpResult(C.Structure):
    _fields_ = [("converged", C.c_int), ("iterations", C.c_int), ("status", C.c_int),
                ("R", C.c_double * 9), ("t", C.c_double * 3), ("icp_cov", C.c_double * 36),
                ("time_ms", C.c_double)]


class TrialResult(C.Structure):
    _fields_ = [("converged", C.c_int), ("iterations", C.c_int), ("status", C.c_int),
                ("time_ms", C.c_double), ("trans_error_m", C.c_double), ("rot_error_deg", C.c_double),
                ("final_rmse", C.c_double), ("final_fitness", C.c_double), ("corr_num", C.c_int64),
                ("final_transform", C.c_double * 16), ("H_upper", C.c_double * 21),
                ("degenerate_mask", C.c_int * 6)]


_STRUCTS = {"dcreg_lin_params": LinParams, "dcreg_lin_out": LinOut, "dcreg_lin_debug": LinDebug,
            "dcreg_index_info": IndexInfo, "dcreg_config": Config, "dcreg_analysis": Analysis,
            "dcreg_iter_log": IterLog, "dcreg_icp_result": IcpResult, "dcreg_trial_result": TrialResult,
            "dcreg_launch_stats": LaunchStats, "dcreg_method_stats": MethodStats}

# every symbol include/dcreg.h and include/dcreg_debug.h declare
REDUCE_FN = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_void_p)      # dcreg_reduce_fn

EXPORTS = [
    "dcreg_backend_create", "dcreg_backend_destroy", "dcreg_last_error", "dcreg_set_stream", "dcreg_set_option",
    "dcreg_set_target", "dcreg_set_target_device", "dcreg_set_source", "dcreg_set_source_device",
    "dcreg_default_lin_params", "dcreg_linearize", "dcreg_linearize_batch", "dcreg_linearize_batch_begin",
    "dcreg_linearize_batch_end", "dcreg_linearize_batch_begin_warm", "dcreg_reserve_warm_states", "dcreg_reset_warm_state", "dcreg_hint_misalignment", "dcreg_linearize_debug", "dcreg_launch_stats_get", "dcreg_knn", "dcreg_kdtree_build", "dcreg_kdtree_info", "dcreg_knn_timed",
    "dcreg_linearize_gated_begin", "dcreg_linearize_gate_open", "dcreg_linearize_gate_abort",
    "dcreg_index_info_get", "dcreg_kernel_time", "dcreg_launch_series", "dcreg_launch_series_passes", "dcreg_team_pass_stamps", "dcreg_roi_info", "dcreg_default_config", "dcreg_analyze_degeneracy", "dcreg_analyze_degeneracy_two_part",
    "dcreg_solve_degenerate_system", "dcreg_unpack_hessian", "dcreg_boxplus", "dcreg_pose6d_to_matrix",
    "dcreg_pose_error", "dcreg_icp_run", "dcreg_icp_run_sharded", "dcreg_icp_run_many", "dcreg_icp_run_euler", "dcreg_icp_run_trials", "dcreg_icp_run_montecarlo", "dcreg_p2p_error", "dcreg_sizeof", "dcreg_version", "dcreg_trial_pose",
    "dcreg_set_host_threads", "dcreg_get_host_threads", "dcreg_comm_unique_id", "dcreg_comm_init", "dcreg_comm_destroy", "dcreg_comm_allgather_sum", "dcreg_icp_run_sharded_rccl",
    "dcreg_montecarlo_job", "dcreg_comm_allgather", "dcreg_comm_info", "dcreg_set_error_message",
]

_lib = None


class DcregError(RuntimeError):
    pass


def load():
    """dlopen the in-tree HIP library.  Fails loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DcregError("libdcreg_hip.so is missing (%s): run `python -c 'import __graft_entry__ as g; g.build()'`; "
                         "there is no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    dp, fp, ip = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int32)
    vp = C.c_void_p
    L.dcreg_backend_create.restype = C.c_int
    L.dcreg_backend_create.argtypes = [C.POINTER(vp), C.c_int]
    L.dcreg_backend_destroy.restype = None
    L.dcreg_backend_destroy.argtypes = [vp]
    L.dcreg_last_error.restype = C.c_char_p
    L.dcreg_last_error.argtypes = [vp]
    L.dcreg_set_stream.argtypes = [vp, vp]
    L.dcreg_set_option.argtypes = [vp, C.c_char_p, C.c_double]
    for name in ("dcreg_set_target", "dcreg_set_target_device"):
        getattr(L, name).argtypes = [vp, vp, C.c_int64, C.c_int64, C.c_double]
    for name in ("dcreg_set_source", "dcreg_set_source_device"):
        getattr(L, name).argtypes = [vp, vp, C.c_int64, C.c_int64]
    L.dcreg_default_lin_params.argtypes = [C.POINTER(LinParams), C.c_double]
    L.dcreg_linearize.argtypes = [vp, dp, dp, C.POINTER(LinParams), C.POINTER(LinOut)]
    L.dcreg_linearize_batch.argtypes = [vp, C.c_int, dp, dp, C.POINTER(LinParams), C.POINTER(LinOut)]
    L.dcreg_linearize_batch_begin.argtypes = [vp, C.c_int, C.c_int, dp, dp, C.POINTER(LinParams)]
    L.dcreg_linearize_batch_begin_warm.argtypes = [vp, C.c_int, C.c_int, dp, dp, ip, C.POINTER(LinParams)]
    L.dcreg_linearize_batch_end.argtypes = [vp, C.c_int, C.POINTER(LinOut)]
    L.dcreg_reserve_warm_states.argtypes = [vp, C.c_int64]
    L.dcreg_reset_warm_state.argtypes = [vp, C.c_int64]
    L.dcreg_hint_misalignment.argtypes = [vp, C.c_double]
    L.dcreg_launch_stats_get.argtypes = [vp, C.POINTER(LaunchStats), C.c_int]
    L.dcreg_linearize_gated_begin.argtypes = [vp, C.c_int, C.POINTER(LinParams)]
    L.dcreg_linearize_gate_open.argtypes = [vp, dp, dp]
    L.dcreg_linearize_gate_abort.argtypes = [vp]
    L.dcreg_linearize_debug.argtypes = [vp, dp, dp, C.POINTER(LinParams), C.POINTER(LinOut), C.POINTER(LinDebug)]
    L.dcreg_knn.argtypes = [vp, fp, C.c_int64, C.c_int64, C.c_int, C.c_double, ip, fp]
    L.dcreg_kdtree_build.argtypes = [vp, C.c_int]
    L.dcreg_kdtree_info.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), dp]
    L.dcreg_knn_timed.argtypes = [vp, fp, C.c_int64, C.c_int64, C.c_int, C.c_double, C.c_int, C.c_int, ip, fp, dp]
    L.dcreg_index_info_get.argtypes = [vp, C.POINTER(IndexInfo)]
    L.dcreg_kernel_time.argtypes = [vp, dp, C.POINTER(C.c_int64), C.c_int]
    L.dcreg_launch_series.argtypes = [vp, dp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int64, C.c_int]
    if hasattr(L, "dcreg_launch_series_passes"):        # (absent from an older build loaded through DCREG_LIB for an A/B: scripts/ab_multi.sh)
        L.dcreg_launch_series_passes.argtypes = [vp, C.POINTER(C.c_uint8), C.c_int64]
    if hasattr(L, "dcreg_team_pass_stamps"):
        L.dcreg_team_pass_stamps.argtypes = [vp, C.POINTER(C.c_uint64), C.c_int64]
    L.dcreg_default_config.restype = None
    L.dcreg_default_config.argtypes = [C.POINTER(Config)]
    L.dcreg_analyze_degeneracy.argtypes = [dp, C.c_int, C.c_int, C.POINTER(Config), C.POINTER(Analysis)]
    L.dcreg_analyze_degeneracy_two_part.argtypes = [dp, C.c_int, C.c_int, C.POINTER(Config), C.POINTER(Analysis), C.POINTER(C.c_int)]
    L.dcreg_solve_degenerate_system.argtypes = [dp, dp, C.c_int, C.POINTER(Config), C.POINTER(Analysis), dp]
    L.dcreg_unpack_hessian.restype = None
    L.dcreg_unpack_hessian.argtypes = [dp, dp]
    L.dcreg_boxplus.restype = None
    L.dcreg_boxplus.argtypes = [dp, dp, dp, dp, dp]
    L.dcreg_pose6d_to_matrix.restype = None
    L.dcreg_pose6d_to_matrix.argtypes = [C.c_double] * 6 + [dp]
    L.dcreg_pose_error.restype = None
    L.dcreg_pose_error.argtypes = [dp, dp, dp, dp]
    L.dcreg_icp_run.argtypes = [vp, dp, dp, C.c_int, C.c_int, C.POINTER(Config), C.POINTER(IterLog), C.c_int,
                                C.POINTER(IcpResult)]
    L.dcreg_icp_run_sharded.argtypes = [vp, dp, dp, C.c_int, C.c_int, C.POINTER(Config), C.c_int64, REDUCE_FN, vp,
                                        C.POINTER(IterLog), C.c_int, C.POINTER(IcpResult)]
    L.dcreg_comm_unique_id.argtypes = [vp]
    L.dcreg_comm_init.argtypes = [vp, vp, C.c_int, C.c_int]
    L.dcreg_comm_destroy.argtypes = [vp]
    L.dcreg_comm_allgather_sum.argtypes = [vp, dp]
    L.dcreg_icp_run_sharded_rccl.argtypes = [vp, dp, dp, C.c_int, C.c_int, C.POINTER(Config), C.c_int64, C.POINTER(IterLog), C.c_int,
                                             C.POINTER(IcpResult)]
    L.dcreg_icp_run_many.argtypes = [C.c_int, C.POINTER(vp), dp, dp, C.c_int, C.c_int, C.POINTER(Config), C.POINTER(IcpResult)]
    L.dcreg_icp_run_euler.argtypes = [vp, dp, C.c_int, C.c_int, C.POINTER(Config), C.POINTER(IterLog), C.c_int,
                                      C.POINTER(IcpResult), dp]
    L.dcreg_icp_run_trials.argtypes = [vp, C.c_int, dp, dp, C.c_int, C.c_int, C.POINTER(Config), C.POINTER(TrialResult)]
    L.dcreg_p2p_error.argtypes = [vp, dp, C.c_double, dp, dp, dp, C.POINTER(C.c_int64)]
    L.dcreg_trial_pose.argtypes = [dp, C.c_uint64, C.c_int64, C.c_double, C.c_double, dp, dp]
    L.dcreg_set_host_threads.argtypes = [C.c_int]
    L.dcreg_get_host_threads.argtypes = []
    L.dcreg_icp_run_montecarlo.argtypes = [vp, dp, C.c_uint64, C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_int, C.c_int,
                                           C.POINTER(Config), C.c_int, C.POINTER(TrialResult)]
    L.dcreg_montecarlo_job.argtypes = [vp, dp, C.c_uint64, C.c_int64, C.c_double, C.c_double, C.c_int, C.c_int, C.POINTER(Config), C.c_int, dp,
                                       C.POINTER(MethodStats)]
    L.dcreg_comm_allgather.argtypes = [vp, dp, dp, C.c_int64]
    L.dcreg_comm_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.dcreg_set_error_message.restype = None
    L.dcreg_set_error_message.argtypes = [vp, C.c_char_p]
    L.dcreg_sizeof.restype = C.c_size_t
    L.dcreg_sizeof.argtypes = [C.c_char_p]
    L.dcreg_version.restype = C.c_char_p
    for name, st in _STRUCTS.items():
        if L.dcreg_sizeof(name.encode()) != C.sizeof(st):
            raise DcregError("struct layout mismatch for %s" % name)
    _lib = L
    return L


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _f64(a, n=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if n is not None:
        a = a.reshape(n)
    return a


def default_config(**kw):
    cfg = Config()
    load().dcreg_default_config(C.byref(cfg))
    for k, v in kw.items():
        if k == "gt_matrix":
            cfg.gt_matrix = (C.c_double * 16)(*_f64(v, 16))
        else:
            if not hasattr(cfg, k):
                raise AttributeError(k)
            setattr(cfg, k, v)
    return cfg


def default_lin_params(search_radius=1.0, use_weight_derivative=0, euler_rpy=None, euler_exact=False):
    """euler_rpy = (roll, pitch, yaw): the Euler row of the second engine (R, t passed to linearize must be the pose6d_matrix of
    that pose) - DCREG_PARAM_EULER = as the reference writes it (icp_test_runner.cpp:2299-2346), euler_exact =
    DCREG_PARAM_EULER_EXACT, the exact roll / pitch / yaw derivative."""
    p = LinParams()
    load().dcreg_default_lin_params(C.byref(p), float(search_radius))
    p.use_weight_derivative = int(use_weight_derivative)
    if euler_rpy is not None:
        p.parameterization = 2 if euler_exact else 1
        p.euler_rpy[:] = [float(v) for v in euler_rpy]
    return p


def unpack_hessian(H_upper):
    H = np.empty(36)
    load().dcreg_unpack_hessian(_dp(_f64(H_upper, 21)), _dp(H))
    return H.reshape(6, 6)


def analyze_degeneracy(H, detection, handling, cfg):
    an = Analysis()
    rc = load().dcreg_analyze_degeneracy(_dp(_f64(H, 36)), DETECTION[detection], HANDLING[handling], C.byref(cfg), C.byref(an))
    if rc:
        raise DcregError("dcreg_analyze_degeneracy rc=%d" % rc)
    return an


def analyze_degeneracy_two_part(H, detection, handling, cfg):
    """the analysis as the pipelined engine takes it (dcreg_debug.h) -> (Analysis, owed mask)"""
    an, owed = Analysis(), C.c_int(0)
    rc = load().dcreg_analyze_degeneracy_two_part(_dp(_f64(H, 36)), DETECTION[detection], HANDLING[handling], C.byref(cfg), C.byref(an), C.byref(owed))
    if rc:
        raise DcregError("dcreg_analyze_degeneracy_two_part rc=%d" % rc)
    return an, owed.value


def solve_degenerate_system(H, g, handling, cfg, an):
    x = np.empty(6)
    rc = load().dcreg_solve_degenerate_system(_dp(_f64(H, 36)), _dp(_f64(g, 6)), HANDLING[handling], C.byref(cfg), C.byref(an), _dp(x))
    if rc:
        raise DcregError("dcreg_solve_degenerate_system rc=%d" % rc)
    return x


def boxplus(R, t, dx):
    Ro, to = np.empty(9), np.empty(3)
    load().dcreg_boxplus(_dp(_f64(R, 9)), _dp(_f64(t, 3)), _dp(_f64(dx, 6)), _dp(Ro), _dp(to))
    return Ro.reshape(3, 3), to


def pose6d_to_matrix(roll, pitch, yaw, x, y, z):
    T = np.empty(16)
    load().dcreg_pose6d_to_matrix(roll, pitch, yaw, x, y, z, _dp(T))
    return T.reshape(4, 4)


def pose_error(gt, T):
    a, b = C.c_double(), C.c_double()
    load().dcreg_pose_error(_dp(_f64(gt, 16)), _dp(_f64(T, 16)), C.byref(a), C.byref(b))
    return a.value, b.value


def trial_pose(base_xyzrpy, seed, k, trans_amp, rot_amp_rad):
    """dcreg_trial_pose: initial pose of Monte-Carlo trial k (k == 0: the base pose), 4x4."""
    T = np.empty(16)
    rc = load().dcreg_trial_pose(_dp(_f64(base_xyzrpy, 6)), int(seed), int(k), float(trans_amp), float(rot_amp_rad), _dp(T), None)
    if rc:
        raise DcregError("dcreg_trial_pose rc=%d" % rc)
    return T.reshape(4, 4)


def set_host_threads(n):
    """dcreg_set_host_threads: OpenMP threads of the batched engines' host steps (overrides a launcher's OMP_NUM_THREADS=1)."""
    if load().dcreg_set_host_threads(int(n)) != 0:
        raise DcregError("dcreg_set_host_threads(%d)" % n)
    return load().dcreg_get_host_threads()


def comm_unique_id():
    """dcreg_comm_unique_id: 128 opaque bytes identifying a new RCCL communicator (rank 0 calls it and shares the bytes)."""
    buf = (C.c_char * 128)()
    rc = load().dcreg_comm_unique_id(C.cast(buf, C.c_void_p))
    if rc:
        raise DcregError("dcreg_comm_unique_id rc=%d (RCCL unavailable?)" % rc)
    return bytes(buf.raw)


def icp_run_many(contexts, T0s, method, cfg):
    """dcreg_icp_run_many: independent scan pairs (one Context each) at once, one host thread per pair."""
    n = len(contexts)
    T0s = _f64(T0s).reshape(n, 4, 4)
    R0 = np.ascontiguousarray(T0s[:, :3, :3]).reshape(n, 9)
    t0 = np.ascontiguousarray(T0s[:, :3, 3]).reshape(n, 3)
    det, hand = METHODS[method] if isinstance(method, str) else method
    handles = (C.c_void_p * max(n, 1))(*[c._h for c in contexts])
    res = (IcpResult * max(n, 1))()
    rc = load().dcreg_icp_run_many(n, handles, _dp(R0), _dp(t0), DETECTION[det], HANDLING[hand], C.byref(cfg), res)
    if rc:
        raise DcregError("dcreg_icp_run_many rc=%d" % rc)
    return [res[i] for i in range(n)]


class Context:
    """One device context (one GPU, one stream).  Stands for ICPContext (utils.hpp:340-425)."""

    def __init__(self, device=0):
        self._L = load()
        self._h = C.c_void_p()
        rc = self._L.dcreg_backend_create(C.byref(self._h), int(device))
        if rc != OK:
            self._h = None
            raise DcregError("dcreg_backend_create(device=%d) failed with %d: no usable HIP device and no CPU fallback" % (device, rc))
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self._L.dcreg_backend_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != OK:
            raise DcregError("%s failed (%d): %s" % (what, rc, self._L.dcreg_last_error(self._h).decode()))

    def set_stream(self, hip_stream_ptr):
        self._check(self._L.dcreg_set_stream(self._h, C.c_void_p(hip_stream_ptr or 0)), "dcreg_set_stream")

    def set_option(self, key, value):
        self._check(self._L.dcreg_set_option(self._h, key.encode(), float(value)), "dcreg_set_option")

    def set_target(self, xyz, search_radius):
        a = np.ascontiguousarray(xyz, dtype=np.float32)
        n, stride = a.shape[0], a.shape[1]
        self._check(self._L.dcreg_set_target(self._h, a.ctypes.data, n, stride, float(search_radius)), "dcreg_set_target")

    def set_source(self, xyz):
        a = np.ascontiguousarray(xyz, dtype=np.float32)
        self._check(self._L.dcreg_set_source(self._h, a.ctypes.data, a.shape[0], a.shape[1]), "dcreg_set_source")

    def set_target_device(self, dev_ptr, n, stride, search_radius):
        self._check(self._L.dcreg_set_target_device(self._h, C.c_void_p(dev_ptr), n, stride, float(search_radius)), "dcreg_set_target_device")

    def set_source_device(self, dev_ptr, n, stride):
        self._check(self._L.dcreg_set_source_device(self._h, C.c_void_p(dev_ptr), n, stride), "dcreg_set_source_device")

    def index_info(self):
        info = IndexInfo()
        self._L.dcreg_index_info_get(self._h, C.byref(info))
        return info

    @staticmethod
    def _out_dict(o):
        d = {"H_upper": np.array(o.H_upper[:]), "g": np.array(o.g[:]), "sum_r2": o.sum_r2, "sum_b2": o.sum_b2,
             "n_eff": o.n_eff, "n_pt": o.n_pt}
        d["H"] = unpack_hessian(d["H_upper"])
        return d

    def linearize(self, R, t, params=None, debug=False):
        params = params or default_lin_params()
        R, t = _f64(R, 9), _f64(t, 3)
        out = LinOut()
        if not debug:
            self._check(self._L.dcreg_linearize(self._h, _dp(R), _dp(t), C.byref(params), C.byref(out)), "dcreg_linearize")
            return self._out_dict(out)
        n = self.index_info().n_source
        keep = {"nn_idx": np.full((n, 5), -1, np.int32), "nn_d2": np.full((n, 5), np.inf, np.float32),
                "flag": np.zeros(n, np.uint8), "normal": np.zeros((n, 3)), "r": np.zeros(n), "s": np.zeros(n),
                "stats": np.zeros(n, np.uint32)}
        dbg = LinDebug(keep["nn_idx"].ctypes.data_as(C.POINTER(C.c_int32)), keep["nn_d2"].ctypes.data_as(C.POINTER(C.c_float)),
                       keep["flag"].ctypes.data_as(C.POINTER(C.c_uint8)), _dp(keep["normal"]), _dp(keep["r"]), _dp(keep["s"]),
                       keep["stats"].ctypes.data_as(C.POINTER(C.c_uint32)))
        self._check(self._L.dcreg_linearize_debug(self._h, _dp(R), _dp(t), C.byref(params), C.byref(out), C.byref(dbg)), "dcreg_linearize_debug")
        d = self._out_dict(out)
        d.update(keep)
        return d

    def linearize_stamped(self, R, t, params=None):
        """timing probe (dcreg_debug.h dcreg_lin_debug::stamps): a plain linearisation (certificates in use) whose waves record shader-clock
        stamps at their phase boundaries -> (sums dict, stamps [n_waves, 8] uint64: t_start, t_loaded, t_searched, t_fitted, t_row, t_reduced,
        lanes searched, lanes refitted; waves that skipped a phase leave its stamp 0)"""
        params = params or default_lin_params()
        R, t = _f64(R, 9), _f64(t, 3)
        out = LinOut()
        n = self.index_info().n_source
        nw = 4 * ((n + 255) // 256)
        st = np.zeros((nw, 8), np.uint64)
        dbg = LinDebug()
        dbg.stamps = st.ctypes.data_as(C.POINTER(C.c_uint64))
        self._check(self._L.dcreg_linearize_debug(self._h, _dp(R), _dp(t), C.byref(params), C.byref(out), C.byref(dbg)), "dcreg_linearize_debug")
        return self._out_dict(out), st

    def linearize_raw(self, R, t, params, out):
        """Hot-loop variant: caller-owned float64 arrays / structs, no allocation."""
        return self._L.dcreg_linearize(self._h, _dp(R), _dp(t), C.byref(params), C.byref(out))

    def linearize_batch(self, Rs, ts, params=None):
        params = params or default_lin_params()
        Rs = _f64(Rs).reshape(-1, 9)
        ts = _f64(ts).reshape(-1, 3)
        n = Rs.shape[0]
        outs = (LinOut * n)()
        self._check(self._L.dcreg_linearize_batch(self._h, n, _dp(Rs), _dp(ts), C.byref(params), outs), "dcreg_linearize_batch")
        return [self._out_dict(o) for o in outs]

    def reserve_warm_states(self, n_states):
        self._check(self._L.dcreg_reserve_warm_states(self._h, int(n_states)), "dcreg_reserve_warm_states")

    def hint_misalignment(self, metres):
        """Scheduling hint (never needed for correctness): expected distance of the source points from the map at the next poses."""
        self._check(self._L.dcreg_hint_misalignment(self._h, float(metres)), "dcreg_hint_misalignment")

    def reset_warm_state(self, state_id):
        self._check(self._L.dcreg_reset_warm_state(self._h, int(state_id)), "dcreg_reset_warm_state")

    def launch_stats(self, reset=False):
        """dcreg_launch_stats_get (dcreg_debug.h): how the linearisations since the last reset were carried out"""
        st = LaunchStats()
        self._check(self._L.dcreg_launch_stats_get(self._h, C.byref(st), int(reset)), "dcreg_launch_stats_get")
        return {k: int(getattr(st, k)) for k, _ in LaunchStats._fields_}

    def linearize_batch_warm(self, Rs, ts, state_ids, params=None, slot=0):
        """dcreg_linearize_batch_begin_warm + _end: pose i reads and updates warm-start state state_ids[i] (-1 = cold)."""
        params = params or default_lin_params()
        Rs = _f64(Rs).reshape(-1, 9)
        ts = _f64(ts).reshape(-1, 3)
        n = Rs.shape[0]
        ids = np.ascontiguousarray(state_ids, dtype=np.int32).reshape(n)
        outs = (LinOut * n)()
        self._check(self._L.dcreg_linearize_batch_begin_warm(self._h, slot, n, _dp(Rs), _dp(ts), ids.ctypes.data_as(C.POINTER(C.c_int32)),
                                                             C.byref(params)), "dcreg_linearize_batch_begin_warm")
        self._check(self._L.dcreg_linearize_batch_end(self._h, slot, outs), "dcreg_linearize_batch_end")
        return [self._out_dict(o) for o in outs]

    # pipelined single-pose launches (what dcreg_icp_run does between two iterations)
    def linearize_begin(self, R, t, params=None, slot=0):
        params = params or default_lin_params()
        R = _f64(R).reshape(9); t = _f64(t).reshape(3)
        self._check(self._L.dcreg_linearize_batch_begin(self._h, slot, 1, _dp(R), _dp(t), C.byref(params)), "dcreg_linearize_batch_begin")

    def linearize_gated_begin(self, params=None, slot=0):
        """queue a linearisation whose pose arrives later (gate_open) or never (gate_abort)"""
        params = params or default_lin_params()
        self._check(self._L.dcreg_linearize_gated_begin(self._h, slot, C.byref(params)), "dcreg_linearize_gated_begin")

    def gate_open(self, R, t):
        R = _f64(R).reshape(9); t = _f64(t).reshape(3)
        self._check(self._L.dcreg_linearize_gate_open(self._h, _dp(R), _dp(t)), "dcreg_linearize_gate_open")

    def gate_abort(self):
        self._check(self._L.dcreg_linearize_gate_abort(self._h), "dcreg_linearize_gate_abort")

    def linearize_end(self, slot=0):
        out = (LinOut * 1)()
        self._check(self._L.dcreg_linearize_batch_end(self._h, slot, out), "dcreg_linearize_batch_end")
        return self._out_dict(out[0])

    def knn(self, q, k=5, max_radius=0.0):
        q = np.ascontiguousarray(q, dtype=np.float32).reshape(-1, 3)
        idx = np.empty((q.shape[0], k), np.int32)
        d2 = np.empty((q.shape[0], k), np.float32)
        self._check(self._L.dcreg_knn(self._h, q.ctypes.data_as(C.POINTER(C.c_float)), q.shape[0], 3, k, float(max_radius),
                                      idx.ctypes.data_as(C.POINTER(C.c_int32)), d2.ctypes.data_as(C.POINTER(C.c_float))), "dcreg_knn")
        return idx, d2

    def kdtree_build(self, leaf_size=16):
        """kd-tree comparator over the current target (dcreg_debug.h); returns (depth, leaf_size, host build ms)"""
        self._check(self._L.dcreg_kdtree_build(self._h, int(leaf_size)), "dcreg_kdtree_build")
        d, l, ms = C.c_int32(), C.c_int32(), C.c_double()
        self._check(self._L.dcreg_kdtree_info(self._h, C.byref(d), C.byref(l), C.byref(ms)), "dcreg_kdtree_info")
        return d.value, l.value, ms.value

    def knn_timed(self, q, k=5, max_radius=0.0, index="grid", repeats=10):
        """exact k-NN on the grid or on the kd-tree comparator -> (idx, d2, kernel ms per launch)"""
        q = np.ascontiguousarray(q, dtype=np.float32).reshape(-1, 3)
        idx = np.empty((q.shape[0], k), np.int32)
        d2 = np.empty((q.shape[0], k), np.float32)
        ms = C.c_double()
        self._check(self._L.dcreg_knn_timed(self._h, q.ctypes.data_as(C.POINTER(C.c_float)), q.shape[0], 3, k, float(max_radius),
                                            {"grid": 0, "kdtree": 1, "grid_sweep": 2}[index], int(repeats), idx.ctypes.data_as(C.POINTER(C.c_int32)),
                                            d2.ctypes.data_as(C.POINTER(C.c_float)), C.byref(ms)), "dcreg_knn_timed")
        return idx, d2, ms.value

    def launch_series(self, reset=True, cap=1 << 20):
        """dcreg_launch_series (dcreg_debug.h; option "record_launches"): per completed launch its HIP-event time (ms, -1 = untimed),
        points searched, points refitted and points linearised, which pass ran in front (dcreg_launch_series_passes: 0 / 1 / 2) and
        whether the linearisation kernel ran in one-wave blocks -> dict of arrays"""
        lp = C.POINTER(C.c_int64)
        n = self._L.dcreg_launch_series(self._h, None, None, None, None, 0, 0)
        if n < 0:
            raise DcregError("dcreg_launch_series failed")
        n = min(n, cap)
        ms = np.empty(n, np.float64)
        se, rf, pt = np.empty(n, np.int64), np.empty(n, np.int64), np.empty(n, np.int64)
        adv = np.zeros(n, np.uint8)
        if hasattr(self._L, "dcreg_launch_series_passes"):
            self._L.dcreg_launch_series_passes(self._h, adv.ctypes.data_as(C.POINTER(C.c_uint8)), n)
        self._L.dcreg_launch_series(self._h, _dp(ms), se.ctypes.data_as(lp), rf.ctypes.data_as(lp), pt.ctypes.data_as(lp), n, int(reset))
        return {"ms": ms, "searched": se, "refitted": rf, "points": pt, "advanced": adv & 3, "one_wave": (adv >> 2) & 1}

    def team_pass_stamps(self):
        """dcreg_team_pass_stamps (option "team_stamps"): [n_blocks, 8] shader-clock words of the last launch that ran the small-frame pass"""
        n = self._L.dcreg_team_pass_stamps(self._h, None, 0)
        if n <= 0:
            return np.zeros((0, 8), np.uint64)
        st = np.zeros((n + 1, 8), np.uint64)          # the last row: outcome counts (served with slack, layers, wide, rows, list, OUT, no slack, refit)
        self._L.dcreg_team_pass_stamps(self._h, st.ctypes.data_as(C.POINTER(C.c_uint64)), n + 1)
        return st

    def roi_info(self):
        """dcreg_roi_info: the window index of a large map (dcreg_debug.h)"""
        v = (C.c_double * 11)()
        self._L.dcreg_roi_info.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        self._check(self._L.dcreg_roi_info(self._h, v), "dcreg_roi_info")
        return {"box_min": [v[0], v[1], v[2]], "box_max": [v[3], v[4], v[5]], "points": int(v[6]), "cell": v[7], "windows_built": int(v[8]),
                "active": bool(v[9]), "whole_map_capped": bool(v[10])}

    def kernel_time(self, reset=False):
        ms, n = C.c_double(), C.c_int64()
        self._L.dcreg_kernel_time(self._h, C.byref(ms), C.byref(n), int(reset))
        return ms.value, n.value

    def icp_run(self, T0, method, cfg, log_capacity=None):
        T0 = _f64(T0).reshape(4, 4)
        R0, t0 = np.ascontiguousarray(T0[:3, :3]).reshape(9), np.ascontiguousarray(T0[:3, 3])
        det, hand = METHODS[method] if isinstance(method, str) else method
        cap = cfg.max_iterations if log_capacity is None else log_capacity
        logs = (IterLog * max(cap, 1))()
        res = IcpResult()
        self._check(self._L.dcreg_icp_run(self._h, _dp(R0), _dp(t0), DETECTION[det], HANDLING[hand], C.byref(cfg), logs, cap,
                                          C.byref(res)), "dcreg_icp_run")
        n = min(res.iterations, cap)
        if res.status == 1:
            n = min(res.iterations - 1, cap)
        return res, [logs[i] for i in range(max(n, 0))]

    def icp_run_sharded(self, T0, method, cfg, n_source_total, reduce_rows, log_capacity=None):
        """dcreg_icp_run_sharded: reduce_rows(numpy float64[32] view) must overwrite the row IN PLACE with the sum over
        all ranks in rank order (see dcreg_amd/pointshard.py)."""
        T0 = _f64(T0).reshape(4, 4)
        R0, t0 = np.ascontiguousarray(T0[:3, :3]).reshape(9), np.ascontiguousarray(T0[:3, 3])
        det, hand = METHODS[method] if isinstance(method, str) else method
        cap = cfg.max_iterations if log_capacity is None else log_capacity
        logs = (IterLog * max(cap, 1))()
        res = IcpResult()
        err = []

        def _cb(row_ptr, _user):
            try:
                reduce_rows(np.ctypeslib.as_array(row_ptr, shape=(32,)))
                return 0
            except Exception as e:          # never let an exception cross the C boundary
                err.append(e)
                return 1
        cb = REDUCE_FN(_cb)
        rc = self._L.dcreg_icp_run_sharded(self._h, _dp(R0), _dp(t0), DETECTION[det], HANDLING[hand], C.byref(cfg),
                                           int(n_source_total), cb, None, logs, cap, C.byref(res))
        if err:
            raise err[0]
        self._check(rc, "dcreg_icp_run_sharded")
        n = min(res.iterations, cap)
        if res.status == 1:
            n = min(res.iterations - 1, cap)
        return res, [logs[i] for i in range(max(n, 0))]

    def comm_init(self, id128, rank, world):
        """dcreg_comm_init: collective (every rank of the job calls it with the same 128 bytes from comm_unique_id())."""
        buf = (C.c_char * 128).from_buffer_copy(bytes(id128))
        self._check(self._L.dcreg_comm_init(self._h, C.cast(buf, C.c_void_p), int(rank), int(world)), "dcreg_comm_init")

    def comm_allgather_sum(self, row):
        row = _f64(row, 32).copy()
        self._check(self._L.dcreg_comm_allgather_sum(self._h, _dp(row)), "dcreg_comm_allgather_sum")
        return row

    def icp_run_sharded_rccl(self, T0, method, cfg, n_source_total, log_capacity=None):
        """dcreg_icp_run_sharded_rccl: point-sharded run, the per-iteration exchange is an RCCL all_gather inside the engine."""
        T0 = _f64(T0).reshape(4, 4)
        R0, t0 = np.ascontiguousarray(T0[:3, :3]).reshape(9), np.ascontiguousarray(T0[:3, 3])
        det, hand = METHODS[method] if isinstance(method, str) else method
        cap = cfg.max_iterations if log_capacity is None else log_capacity
        logs = (IterLog * max(cap, 1))()
        res = IcpResult()
        self._check(self._L.dcreg_icp_run_sharded_rccl(self._h, _dp(R0), _dp(t0), DETECTION[det], HANDLING[hand], C.byref(cfg),
                                                       int(n_source_total), logs, cap, C.byref(res)), "dcreg_icp_run_sharded_rccl")
        n = min(res.iterations, cap)
        if res.status == 1:
            n = min(res.iterations - 1, cap)
        return res, [logs[i] for i in range(max(n, 0))]

    def icp_run_euler(self, pose6d, method, cfg, log_capacity=None):
        """Second engine (Pose6D state, LOAM Jacobian); pose6d = (roll, pitch, yaw, x, y, z).
        Returns (result, [IterLog...], final_pose6d)."""
        p0 = _f64(pose6d, 6)
        det, hand = METHODS[method] if isinstance(method, str) else method
        cap = cfg.max_iterations if log_capacity is None else log_capacity
        logs = (IterLog * max(cap, 1))()
        res = IcpResult()
        pf = np.zeros(6)
        self._check(self._L.dcreg_icp_run_euler(self._h, _dp(p0), DETECTION[det], HANDLING[hand], C.byref(cfg), logs, cap,
                                                C.byref(res), _dp(pf)), "dcreg_icp_run_euler")
        return res, [logs[i] for i in range(max(min(res.iterations, cap), 0))], pf

    def icp_run_montecarlo(self, base_xyzrpy, seed, first_trial, trial_stride, n_trials, trans_amp, rot_amp_rad, method, cfg, slots=0):
        """dcreg_icp_run_montecarlo: this rank's share of the Monte-Carlo experiment, poses generated and trials batched in C++."""
        det, hand = METHODS[method]
        res = (TrialResult * max(int(n_trials), 1))()
        self._check(self._L.dcreg_icp_run_montecarlo(self._h, _dp(_f64(base_xyzrpy, 6)), int(seed), int(first_trial), int(trial_stride), int(n_trials),
                                                     float(trans_amp), float(rot_amp_rad), DETECTION[det], HANDLING[hand], C.byref(cfg), int(slots), res),
                    "dcreg_icp_run_montecarlo")
        return [res[i] for i in range(int(n_trials))]

    def montecarlo_job(self, base_xyzrpy, seed, n_trials, trans_amp, rot_amp_rad, method, cfg, slots=0, want_records=True):
        """dcreg_montecarlo_job: the whole experiment as one C-ABI job over the ranks of this context's communicator (comm_init; none = one
        rank): shard, run, ONE ncclAllGather, statistics - on every rank.  -> (records [n_trials, 64] float64 or None, stats dict)"""
        det, hand = METHODS[method]
        n = int(n_trials)
        rec = np.zeros((max(n, 1), 64)) if want_records else None
        st = MethodStats()
        self._check(self._L.dcreg_montecarlo_job(self._h, _dp(_f64(base_xyzrpy, 6)), int(seed), n, float(trans_amp), float(rot_amp_rad), DETECTION[det],
                                                 HANDLING[hand], C.byref(cfg), int(slots), _dp(rec) if want_records else None, C.byref(st)),
                    "dcreg_montecarlo_job")
        return (rec[:n] if want_records else None), {k: getattr(st, k) for k, _ in MethodStats._fields_}

    def comm_allgather(self, row):
        """dcreg_comm_allgather: this rank's doubles -> [world, len(row)] on every rank"""
        rank, world = C.c_int(), C.c_int()
        self._L.dcreg_comm_info(self._h, C.byref(rank), C.byref(world))
        row = _f64(row).reshape(-1)
        out = np.zeros((world.value, len(row)))
        self._check(self._L.dcreg_comm_allgather(self._h, _dp(row), _dp(out), len(row)), "dcreg_comm_allgather")
        return out

    def icp_run_trials(self, T0s, method, cfg):
        T0s = _f64(T0s).reshape(-1, 4, 4)
        n = T0s.shape[0]
        R0 = np.ascontiguousarray(T0s[:, :3, :3]).reshape(n, 9)
        t0 = np.ascontiguousarray(T0s[:, :3, 3]).reshape(n, 3)
        det, hand = METHODS[method] if isinstance(method, str) else method
        res = (TrialResult * max(n, 1))()
        self._check(self._L.dcreg_icp_run_trials(self._h, n, _dp(R0), _dp(t0), DETECTION[det], HANDLING[hand], C.byref(cfg), res),
                    "dcreg_icp_run_trials")
        return [res[i] for i in range(n)]

    def p2p_error(self, T, error_threshold):
        r, f, ch = C.c_double(), C.c_double(), C.c_double()
        v = C.c_int64()
        self._check(self._L.dcreg_p2p_error(self._h, _dp(_f64(T, 16)), float(error_threshold), C.byref(r), C.byref(f), C.byref(ch),
                                            C.byref(v)), "dcreg_p2p_error")
        return r.value, f.value, ch.value, v.value
