"""Monte-Carlo / multi-scan-pair driver: independent ICP trials sharded over the GPUs of one node.

The reference's skeleton is TestRunner::runMethod (DCReg/src/icp_test_runner.cpp:331-390: the num_runs loop)
plus updateStatistics / finalizeStatistics (:604-664).  The reference has no RNG (every run is identical);
the seeded perturbation of the initial pose is this build's own definition:

    trial k :  initial_noise = base + U(-a, a) per DoF,  drawn from MT19937(seed + k)

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
    rng = np.random.Generator(np.random.MT19937(int(seed) + int(k)))
    u = rng.uniform(-1.0, 1.0, 6)
    p = np.asarray(base, np.float64) + np.concatenate([u[:3] * trans_amp, u[3:] * rot_amp_rad])
    return pose6d_matrix(*p)


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
