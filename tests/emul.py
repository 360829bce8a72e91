"""ctypes binding of the HOST REPLAY of the device algorithm (tests/host_emul/: dcreg_amd/csrc/device/search.hpp compiled
for the CPU with a small shim).  TEST INFRASTRUCTURE ONLY, like oracle/: it lets the CPU suite check the search logic and
the plane fit of the HIP path against the oracle, and feeds the wave cost model used when designing the search.  Nothing
under dcreg_amd/ imports it."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_emul")
_LIB = os.path.join(_HERE, "libdcreg_emul.so")
_SRC = [os.path.join(_HERE, "emul.cpp"), os.path.join(_HERE, "host_emul_shim.hpp"),
        os.path.join(os.path.dirname(_HERE), "..", "dcreg_amd", "csrc", "device", "search.hpp")]
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


class LinParams(C.Structure):
    _fields_ = [("search_radius", C.c_double), ("max_plane_thickness_sq", C.c_double), ("min_normal_norm", C.c_double),
                ("weight_slope", C.c_double), ("weight_min", C.c_double), ("use_weight_derivative", C.c_int32),
                ("fast_plane_fit", C.c_int32), ("cert_margin", C.c_double), ("cert_inflate", C.c_double)]


def build(force=False):
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < max(os.path.getmtime(f) for f in _SRC):
        subprocess.check_call([CLANG, "-x", "c++", "-std=c++17", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-I" + _HERE,
                               _SRC[0], "-o", _LIB])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.emu_index_build.restype = C.c_void_p
        L.emu_index_build.argtypes = [C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int]
        L.emu_index_free.argtypes = [C.c_void_p]
        L.emu_index_set_sweep.argtypes = [C.c_void_p, C.c_int32]
        L.emu_index_set_team.argtypes = [C.c_void_p, C.c_int32]
        L.emu_index_team_served.restype = C.c_int64
        L.emu_index_team_served.argtypes = [C.c_void_p]
        L.emu_index_info.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
        L.emu_hilbert_order.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        L.emu_linearize.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.POINTER(LinParams),
                                    C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 8 + [C.c_void_p, C.c_int64, C.c_void_p]
        L.emu_plane_fit.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.emu_ball_query.restype = C.c_int64
        L.emu_ball_query.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
        L.emu_knn.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def hilbert_order(xyz):
    xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
    order = np.empty(len(xyz), np.uint32)
    lib().emu_hilbert_order(_ptr(xyz), len(xyz), _ptr(order))
    return order


def plane_fit(Q, fast):
    Q = np.ascontiguousarray(Q, np.float64).reshape(5, 3)
    x = np.empty(3)
    lib().emu_plane_fit(_ptr(Q), int(fast), _ptr(x))
    return x


CERT_MARGIN = 0.05      # dcreg_ctx::opt_cert_margin
CERT_MOVE = 0.5         # dcreg_ctx::opt_cert_move
CERT_INFLATE = 0.005    # dcreg_ctx::opt_cert_inflate


class Index:
    """The device's grid index over a target cloud, built on the host."""

    def __init__(self, xyz, radius, cell=0.0, cell_factor=2.0, gap_field=True, x_subdiv=8, cert_margin=CERT_MARGIN):
        """gap_field: True / False = with / without the empty-space distance field."""
        self.xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        self.radius = radius
        self.cert_margin = cert_margin
        # the cells are sized for the search radius R (1 + margin), as context.hip set_target
        self.ptr = lib().emu_index_build(_ptr(self.xyz), len(self.xyz), float(radius) * (1.0 + cert_margin), float(cell), float(cell_factor), int(gap_field), int(x_subdiv))
        h, dims, nc, gc = C.c_double(), (C.c_int32 * 3)(), C.c_int64(), C.c_int32()
        lib().emu_index_info(self.ptr, C.byref(h), dims, C.byref(nc), C.byref(gc))
        self.cell, self.dims, self.n_cells, self.gap_cap = h.value, tuple(dims), nc.value, gc.value

    def set_sweep(self, on):
        """The searches of linearize(): True = row sweep (what k_lin runs), False = ring walk (-DDCREG_RING_WALK builds, dcreg_knn)."""
        lib().emu_index_set_sweep(self.ptr, int(bool(on)))

    def set_team(self, on):
        """True: the queries a sparse wave of k_lin would hand to team_search6 (warm bound inside the 27-cell block) are searched by the
        scalar replay of that algorithm instead of the lock-step search (team_served() counts them)."""
        lib().emu_index_set_team(self.ptr, int(bool(on)))

    def team_served(self):
        return int(lib().emu_index_team_served(self.ptr))

    def __del__(self):
        if getattr(self, "ptr", None):
            lib().emu_index_free(self.ptr)
            self.ptr = None


def knn(index, q, k=5, max_radius=0.0):
    """Exact k-NN of arbitrary queries through the device search functions (as dcreg_knn / k_knn)."""
    q = np.ascontiguousarray(q, np.float32).reshape(-1, 3)
    idx = np.empty((len(q), k), np.int32)
    d2 = np.empty((len(q), k), np.float32)
    lib().emu_knn(index.ptr, _ptr(q), len(q), int(k), float(max_radius), _ptr(idx), _ptr(d2))
    return idx, d2


def ball_query(index, q, bound):
    """The row enumeration of the small-frame advance pass (kernels.hpp k_advance_team) for one query and squared bound, replayed on the
    host -> (n_inside or a negative code where the device would leave the query to k_lin, idx[7], d2[7]: the seven nearest below the bound)"""
    q = np.ascontiguousarray(q, np.float32).reshape(3)
    idx = np.full(7, -1, np.int32); d2 = np.full(7, np.inf, np.float32)
    n = lib().emu_ball_query(index.ptr, _ptr(q), float(bound), _ptr(idx), _ptr(d2))
    return int(n), idx, d2


class Source:
    """A source cloud in the device's processing order (Hilbert curve of the body-frame position) + its neighbour state."""

    def __init__(self, xyz, hilbert=True):
        xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        self.order = hilbert_order(xyz) if hilbert else np.arange(len(xyz), dtype=np.uint32)
        self.sorted = np.ascontiguousarray(xyz[self.order])
        self.n = len(xyz)
        self.stride = (self.n + 63) & ~63
        self.radius_body = float(np.sqrt((np.abs(xyz).max(axis=0).astype(np.float64) ** 2).sum())) if len(xyz) else 0.0
        self.reset_warm()

    def reset_warm(self):
        """state = 6 neighbour positions + certificate + search position per query (search.hpp kStateRows), the pose of the last launch, the index"""
        self.state, self.prev_pose, self.prev_index = None, None, None
        self.searched = 0          # queries searched by the last linearisation


def linearize(index, source, R, t, radius=None, wd=0, fast=True, warm=True, debug=False, stats=False, trace_cap=0, cert_move=CERT_MOVE,
              plan=None, cert_inflate=CERT_INFLATE):
    """One linearisation through the device functions on the host -> dict like Context.linearize (+ "stats" [n, 8] in
    processing order: candidates, outermost shell, table loads, rows, runs, trips, faces, face skips).
    plan: None = as context.hip decides ("full" on a fresh state / debug / a pose change that may move a point farther than cert_move
    cells, else "cert"), or force "full" / "cert" (a fresh state is always searched in full)."""
    radius = index.radius if radius is None else radius
    prm = LinParams(radius, 0.2 * 0.2, 1e-6, 0.9, 0.1, int(wd), int(fast), index.cert_margin, cert_inflate)
    n = source.n
    fresh = source.state is None or source.prev_index is not index      # a new target voids positions and certificates
    if warm and fresh:
        source.state = np.full(19 * source.stride, 0xA5A5A5A5, np.uint32)      # garbage on purpose: a fresh state is never read
        source.prev_pose, source.prev_index = None, index
    state = source.state if warm else None
    R = np.ascontiguousarray(R, np.float64).reshape(9)
    t = np.ascontiguousarray(t, np.float64).reshape(3)
    pose = np.concatenate([R, t])
    delta, certify = None, 0
    if warm and not fresh and source.prev_pose is not None:
        delta = pose - source.prev_pose
        max_move = np.sqrt((delta[:9] ** 2).sum()) * source.radius_body + np.sqrt((delta[9:] ** 2).sum())
        certify = int(cert_move > 0 and max_move <= cert_move * index.cell and not debug) if plan is None else int(plan == "cert")
    out = np.zeros(32)
    keep = {}
    if debug:
        keep = {"nn_idx": np.full((n, 5), -1, np.int32), "nn_d2": np.full((n, 5), np.inf, np.float32), "flag": np.zeros(n, np.uint8),
                "normal": np.zeros((n, 3)), "r": np.zeros(n), "s": np.zeros(n)}
    st = np.zeros((n, 8), np.uint32) if stats else None
    tr = np.zeros((n, trace_cap), np.uint32) if trace_cap else None
    counts = np.zeros(2, np.int64)
    rc = lib().emu_linearize(index.ptr, _ptr(source.sorted), _ptr(source.order), n, _ptr(R), _ptr(t), C.byref(prm), _ptr(state), source.stride,
                             int(fresh), certify, int(bool(warm)),
                             _ptr(out), _ptr(keep.get("nn_idx")), _ptr(keep.get("nn_d2")), _ptr(keep.get("flag")), _ptr(keep.get("normal")),
                             _ptr(keep.get("r")), _ptr(keep.get("s")), _ptr(st), _ptr(tr), int(trace_cap), _ptr(counts))
    assert rc == 0
    if warm:
        source.prev_pose = pose
    source.searched = int(counts[0])
    res = {"H_upper": out[:21].copy(), "g": out[21:27].copy(), "sum_r2": out[27], "sum_b2": out[28], "n_eff": int(round(out[29])),
           "n_pt": int(round(out[30])), "searched": int(counts[0]), "fitted": int(counts[1]), "plan": "cert" if certify else "full"}
    res.update(keep)
    if stats:
        res["stats"] = st
    if trace_cap:
        res["trace"] = tr          # per query (processing order): (key, trips) pairs of the ring walk's scans, last word = words used
    return res


def wave_cost(stats, width=64):
    """Wave-synchronous cost model: a wave runs every loop to its slowest lane.  Returns per-wave maxima [n_waves, 8]."""
    n = len(stats)
    pad = (-n) % width
    s = np.concatenate([stats, np.zeros((pad, stats.shape[1]), stats.dtype)]) if pad else stats
    return s.reshape(-1, width, stats.shape[1]).max(axis=1)
