"""Monte-Carlo / multi-scan-pair driver: independent ICP trials sharded over the GPUs of one node.

The reference's skeleton is TestRunner::runMethod (DCReg/src/icp_test_runner.cpp:331-390: the num_runs loop)
plus updateStatistics / finalizeStatistics (:604-664).  The reference has no RNG (every run is identical);
the seeded perturbation of the initial pose is this build's own definition:

    trial 0 :  the base pose (the reference's deterministic run)
    trial k :  initial_noise = base + U(-a, a) per DoF,  drawn from MT19937(seed + k)      (k >= 1)

ONE generator for every driver: dcreg_trial_pose of the C-ABI (include/dcreg.h), which the C++ runner calls too.

Trials are embarrassingly parallel: rank r runs trials k = r, r + world, ... on its own GPU (its own copy of
the clouds and index); the only exchange is ONE all_gather of fixed-size per-trial records at the end
(torch.distributed: backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in the CPU tests).
"""
import numpy as np

REC = 64  # doubles per trial record
# record layout
R_CONV, R_ITERS, R_TIME, R_TERR, R_RERR, R_RMSE, R_FIT, R_CORR, R_STATUS, R_TRIAL = range(10)
R_T = 10          # 16 doubles: final transform, row-major
R_H = 26          # 21 doubles: last Hessian, upper triangle
R_MASK = 47       # 6 doubles: degenerate mask


def pose6d_matrix(x, y, z, roll, pitch, yaw):
    """Translation * Rz(yaw) * Ry(pitch) * Rx(roll)  (DCReg/include/utils.hpp:452-460)."""
    cr, sr, cp, sp, cy, sy = np.cos(roll), np.sin(roll), np.cos(pitch), np.sin(pitch), np.cos(yaw), np.sin(yaw)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = [x, y, z]
    return T


def trial_pose(base, k, seed, trans_amp, rot_amp_rad):
    """Initial pose of trial k.  base = (x, y, z, roll, pitch, yaw) [m, rad]."""
    from . import api
    return api.trial_pose(base, seed, k, trans_amp, rot_amp_rad)


def shard_indices(n_trials, rank, world):
    return np.arange(rank, n_trials, world, dtype=np.int64)


def trial_record(k, tr):
    """Pack a dcreg_trial_result (ctypes struct or any object with the same fields) into REC doubles."""
    r = np.zeros(REC)
    r[R_CONV], r[R_ITERS], r[R_TIME] = tr.converged, tr.iterations, tr.time_ms
    r[R_TERR], r[R_RERR], r[R_RMSE], r[R_FIT] = tr.trans_error_m, tr.rot_error_deg, tr.final_rmse, tr.final_fitness
    r[R_CORR], r[R_STATUS], r[R_TRIAL] = tr.corr_num, tr.status, k
    r[R_T:R_T + 16] = np.asarray(tr.final_transform[:], np.float64).reshape(16)
    r[R_H:R_H + 21] = np.asarray(tr.H_upper[:], np.float64)
    r[R_MASK:R_MASK + 6] = np.asarray(tr.degenerate_mask[:], np.float64)
    return r


def gather_records(local, n_trials, dist=None, device="cpu"):
    """all_gather the per-rank record blocks -> [n_trials, REC] ordered by trial id (on every rank)."""
    import torch
    local = np.ascontiguousarray(local, np.float64).reshape(-1, REC)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        out = local
    else:
        world = dist.get_world_size()
        per = (n_trials + world - 1) // world                    # fixed-size blocks: pad with trial id -1
        blk = np.full((per, REC), 0.0)
        blk[:, R_TRIAL] = -1.0
        blk[:local.shape[0]] = local
        t = torch.from_numpy(blk).to(device)
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t)
        out = torch.cat(parts).cpu().numpy()
    out = out[out[:, R_TRIAL] >= 0]
    order = np.argsort(out[:, R_TRIAL], kind="stable")
    out = out[order]
    if out.shape[0] != n_trials:
        raise RuntimeError("gathered %d trial records, expected %d" % (out.shape[0], n_trials))
    return out


def method_statistics(records):
    """updateStatistics + finalizeStatistics (icp_test_runner.cpp:604-664): means, population std, min/max."""
    n = records.shape[0]
    st = {"total_runs": int(n), "converged_runs": int(records[:, R_CONV].sum())}
    if n == 0:
        return st
    st["success_rate"] = st["converged_runs"] / n
    for name, col in (("trans_error", R_TERR), ("rot_error", R_RERR), ("time_ms", R_TIME)):
        v = records[:, col]
        st["mean_" + name] = float(v.sum() / n)
        st["std_" + name] = float(np.sqrt(((v - v.sum() / n) ** 2).sum() / n))     # population std (:660-662)
    st["min_trans_error"], st["max_trans_error"] = float(records[:, R_TERR].min()), float(records[:, R_TERR].max())
    st["min_rot_error"], st["max_rot_error"] = float(records[:, R_RERR].min()), float(records[:, R_RERR].max())
    st["mean_iterations"] = float(records[:, R_ITERS].mean())
    st["mean_rmse"] = float(records[:, R_RMSE].mean())
    st["mean_fitness"] = float(records[:, R_FIT].mean())
    st["corr_num"] = int(records[:, R_CORR].sum())
    return st


def run_montecarlo(run_trials, base_pose, n_trials, seed, trans_amp, rot_amp_rad, rank=0, world=1, dist=None,
                   device="cpu", batch=256):
    """run_trials(T0s [m,4,4]) -> list of trial results (same order).  Returns (records [n_trials, REC], stats)."""
    mine = shard_indices(n_trials, rank, world)
    recs = []
    for b0 in range(0, len(mine), batch):
        ks = mine[b0:b0 + batch]
        T0s = np.stack([trial_pose(base_pose, k, seed, trans_amp, rot_amp_rad) for k in ks]) if len(ks) else np.zeros((0, 4, 4))
        res = run_trials(T0s) if len(ks) else []
        recs += [trial_record(int(k), tr) for k, tr in zip(ks, res)]
    local = np.stack(recs) if recs else np.zeros((0, REC))
    allr = gather_records(local, n_trials, dist, device)
    return allr, method_statistics(allr)


def records_from_results(ks, results):
    """[len(ks), REC] records from a sequence / ctypes array of dcreg_trial_result, without a Python loop over the fields."""
    from . import api
    n = len(ks)
    if n == 0:
        return np.zeros((0, REC))
    arr = results if isinstance(results, np.ndarray) else np.frombuffer((api.TrialResult * n)(*results), dtype=np.dtype(api.TrialResult))
    r = np.zeros((n, REC))
    r[:, R_CONV], r[:, R_ITERS], r[:, R_TIME] = arr["converged"], arr["iterations"], arr["time_ms"]
    r[:, R_TERR], r[:, R_RERR], r[:, R_RMSE], r[:, R_FIT] = arr["trans_error_m"], arr["rot_error_deg"], arr["final_rmse"], arr["final_fitness"]
    r[:, R_CORR], r[:, R_STATUS], r[:, R_TRIAL] = arr["corr_num"], arr["status"], np.asarray(ks, np.float64)
    r[:, R_T:R_T + 16] = arr["final_transform"].reshape(n, 16)
    r[:, R_H:R_H + 21] = arr["H_upper"].reshape(n, 21)
    r[:, R_MASK:R_MASK + 6] = arr["degenerate_mask"].reshape(n, 6)
    return r


def run_montecarlo_native(ctx, method, cfg, base_pose, n_trials, seed, trans_amp, rot_amp_rad, rank=0, world=1, dist=None, device="cpu",
                          slots=0):
    """The same experiment with this rank's whole share run by ONE call into the C++ engine (dcreg_icp_run_montecarlo: poses generated
    there, trials batched continuously).  Returns (records [n_trials, REC] on every rank, stats)."""
    mine = shard_indices(n_trials, rank, world)
    res = ctx.icp_run_montecarlo(base_pose, seed, rank, world, len(mine), trans_amp, rot_amp_rad, method, cfg, slots=slots)
    local = records_from_results(mine, res)
    allr = gather_records(local, n_trials, dist, device)
    return allr, method_statistics(allr)


def main(argv=None):
    """The Monte-Carlo experiment end to end (BASELINE config 5): `python -m dcreg_amd.montecarlo --trials 5000` on one GPU, or
    `python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m dcreg_amd.montecarlo --trials 5000`
    (one rank per GPU, trials k = rank mod world, one RCCL all_gather of the trial records at the end)."""
    import argparse, json, os, time
    ap = argparse.ArgumentParser(description=main.__doc__)
    from .scenes import FIXTURE_PCD
    ap.add_argument("--cloud", default=FIXTURE_PCD,
                    help="PCD (binary or ascii, float32 x y z ...) used as source AND target, like the reference's simulated experiment")
    ap.add_argument("--trials", type=int, default=5000)
    ap.add_argument("--methods", default="Ours,ME-SR,ME-TSVD,ME-TReg,FCN-SR")
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--trans-amp", type=float, default=0.5, help="uniform perturbation amplitude per translation DoF [m]")
    ap.add_argument("--rot-amp-deg", type=float, default=2.0)
    ap.add_argument("--base", default="0.2,0.8,0.5,0.1,0.1,2.0", help="base initial pose x,y,z [m], roll,pitch,yaw [deg] (paper run)")
    ap.add_argument("--radius", type=float, default=1.0)
    ap.add_argument("--max-iterations", type=int, default=30)
    ap.add_argument("--batch", type=int, default=256, help="trials in flight per GPU")
    ap.add_argument("--host-threads", type=int, default=0, help="OpenMP threads of this rank's host steps (0 = its share of the usable CPUs)")
    ap.add_argument("--out", default=None, help="write {method: statistics} as JSON (rank 0)")
    a = ap.parse_args(argv)

    import torch
    from . import api, Context, hostinfo
    rank, world, local = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    # torch.distributed.run exports OMP_NUM_THREADS=1; the engine's host steps want this rank's share of the CPUs (computed before
    # the rank is pinned to its slice)
    share, _ = hostinfo.setup_rank(local, local_world)
    api.set_host_threads(a.host_threads if a.host_threads > 0 else share)
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    pts = _read_pcd_xyz(a.cloud)
    ctx = Context(local)
    ctx.set_target(pts, a.radius); ctx.set_source(pts)
    b = [float(v) for v in a.base.split(",")]
    base = (b[0], b[1], b[2], np.deg2rad(b[3]), np.deg2rad(b[4]), np.deg2rad(b[5]))
    cfg = api.default_config(search_radius=a.radius, max_iterations=a.max_iterations, CONVERGENCE_THRESH_TRANS=1e-3, CONVERGENCE_THRESH_ROT=1e-5,
                             KAPPA_TARGET=10.0, STD_REG_GAMMA=100.0, DEGENERACY_THRES_COND=10.0, DEGENERACY_THRES_EIG=120.0,
                             use_weight_derivative=1, always_compute_schur=1)
    out = {}
    for method in a.methods.split(","):
        t0 = time.perf_counter()
        recs, stats = run_montecarlo_native(ctx, method, cfg, base, a.trials, a.seed, a.trans_amp, np.deg2rad(a.rot_amp_deg), rank=rank,
                                            world=world, dist=dist, device="cuda" if world > 1 else "cpu", slots=a.batch)
        el = time.perf_counter() - t0
        stats["wall_s"] = el
        stats["icp_iterations_per_s"] = float(recs[:, R_ITERS].sum()) / el
        out[method] = stats
        if rank == 0:
            print("%-8s %d trials on %d GPU(s): success %.1f %%, trans %.4f +- %.4f m, rot %.4f +- %.4f deg, %.1f iterations/trial, %.2f s (%.0f ICP iterations/s)" % (
                method, a.trials, world, 100 * stats["success_rate"], stats["mean_trans_error"], stats["std_trans_error"], stats["mean_rot_error"],
                stats["std_rot_error"], stats["mean_iterations"], el, stats["icp_iterations_per_s"]), flush=True)
    if rank == 0 and a.out:
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return out


def _read_pcd_xyz(path):
    """Minimal PCD v0.7 reader (binary / ascii, 4-byte fields) -> float32 [n, 3]."""
    with open(path, "rb") as f:
        fields, sizes, counts, n, kind = [], [], [], 0, None
        while True:
            line = f.readline().decode("ascii", "replace").strip()
            if line.startswith("FIELDS"): fields = line.split()[1:]
            elif line.startswith("SIZE"): sizes = [int(v) for v in line.split()[1:]]
            elif line.startswith("COUNT"): counts = [int(v) for v in line.split()[1:]]
            elif line.startswith("POINTS"): n = int(line.split()[1])
            elif line.startswith("DATA"):
                kind = line.split()[1]
                break
        counts = counts or [1] * len(fields)
        if kind == "ascii":
            arr = np.loadtxt(f, dtype=np.float64).reshape(n, -1)
            return np.ascontiguousarray(arr[:, [fields.index(c) for c in "xyz"]], dtype=np.float32)
        if kind != "binary" or any(s != 4 for s in sizes):
            raise ValueError("unsupported PCD layout in %s" % path)
        width = sum(counts)
        raw = np.frombuffer(f.read(n * width * 4), dtype=np.float32).reshape(n, width)
        off = np.cumsum([0] + counts)
        return np.ascontiguousarray(raw[:, [off[fields.index(c)] for c in "xyz"]])


if __name__ == "__main__":
    main()
