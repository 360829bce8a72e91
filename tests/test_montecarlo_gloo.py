"""Multi-process (gloo, world_size 2) test of the trial sharding + statistics gather used for the multi-GPU
Monte-Carlo path.  On CPU the per-trial ICP is played by the oracle (tests may use it); on the GPU node the
same driver is fed by Context.icp_run_trials and the gather runs over RCCL."""
import os
import socket
import types

import numpy as np
import pytest

import helpers as h
from dcreg_amd import montecarlo as mc

N_TRIALS = 7
BASE = (0.2, 0.8, 0.5, h.deg2rad(0.1), h.deg2rad(0.1), h.deg2rad(2.0))


def _oracle_runner():
    from oracle import pyoracle as po
    pts = h.cylinder_cloud()
    tree = po.KdTree(pts)
    cfg = po.default_config(search_radius=1.0, max_iterations=12, thresh_trans=1e-3, thresh_rot=1e-5, kappa_target=10.0,
                            std_reg_gamma=100.0, use_weight_derivative=1, always_compute_schur=1, num_threads=2)

    def run(T0s):
        out = []
        for T0 in T0s:
            res, logs = po.icp_run(tree, pts, T0, "Ours", cfg)
            T = np.eye(4); T[:3, :3] = np.array(res.R[:]).reshape(3, 3); T[:3, 3] = res.t[:]
            te, re_ = po.pose_error(np.eye(4), T)
            last = logs[-1] if logs else None
            out.append(types.SimpleNamespace(
                converged=res.converged, iterations=res.iterations, status=res.status, time_ms=1.0, trans_error_m=te,
                rot_error_deg=re_, final_rmse=last.rmse if last else 0.0, final_fitness=last.fitness if last else 0.0,
                corr_num=last.n_eff if last else 0, final_transform=T.reshape(16), H_upper=np.array(last.H_upper[:]) if last else np.zeros(21),
                degenerate_mask=np.array(last.an.mask[:]) if last else np.zeros(6)))
        return out
    return run


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    recs, stats = mc.run_montecarlo(_oracle_runner(), BASE, N_TRIALS, seed=123, trans_amp=0.3, rot_amp_rad=h.deg2rad(1.0),
                                    rank=rank, world=world, dist=dist, batch=2)
    q.put((rank, recs, stats))
    dist.barrier()
    dist.destroy_process_group()


def test_trial_poses_are_seeded_and_rank_independent():
    a = mc.trial_pose(BASE, 5, 123, 0.3, 0.01)
    b = mc.trial_pose(BASE, 5, 123, 0.3, 0.01)
    c = mc.trial_pose(BASE, 6, 123, 0.3, 0.01)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    assert np.allclose(a[:3, :3] @ a[:3, :3].T, np.eye(3), atol=1e-14)
    assert np.max(np.abs(a[:3, 3] - np.array(BASE[:3]))) <= 0.3
    idx = [mc.shard_indices(10, r, 4) for r in range(4)]
    assert sorted(np.concatenate(idx).tolist()) == list(range(10))
    assert np.allclose(mc.pose6d_matrix(*BASE), h.pose6d_matrix(*BASE[:3], *BASE[3:]))


def test_one_generator_for_every_driver():
    """dcreg_trial_pose is THE definition (the C++ runner and montecarlo.py both call it): trial 0 is the base pose, trial k
    draws six 53-bit doubles from MT19937 seeded with the low 32 bits of seed + k -- numpy's RandomState is the same
    generator, so the values are pinned here independently of the C++ implementation."""
    from dcreg_amd import api
    assert np.array_equal(api.trial_pose(BASE, 123, 0, 0.3, 0.01), mc.pose6d_matrix(*BASE))
    for seed, k in ((123, 1), (2024, 4999), (2 ** 40 + 3, 12)):
        u = np.random.RandomState((seed + k) & 0xFFFFFFFF).random_sample(6) * 2.0 - 1.0
        want = mc.pose6d_matrix(*(np.array(BASE) + np.concatenate([u[:3] * 0.3, u[3:] * 0.01])))
        assert np.allclose(api.trial_pose(BASE, seed, k, 0.3, 0.01), want, rtol=0, atol=1e-15)
    # first two uniforms of MT19937(1) as 53-bit doubles (known-answer: 0.417022004702574, 0.7203244934421581)
    T = api.trial_pose((0, 0, 0, 0, 0, 0), 0, 1, 1.0, 0.0)
    assert np.allclose(T[:3, 3][:2], [2 * 0.417022004702574 - 1, 2 * 0.7203244934421581 - 1], rtol=0, atol=1e-15)


def test_statistics_definitions():
    recs = np.zeros((4, mc.REC))
    recs[:, mc.R_CONV] = [1, 0, 1, 1]
    recs[:, mc.R_TERR] = [0.1, 0.2, 0.3, 0.4]
    recs[:, mc.R_RERR] = [1.0, 2.0, 3.0, 4.0]
    recs[:, mc.R_TIME] = [5, 5, 5, 5]
    recs[:, mc.R_ITERS] = [10, 30, 8, 12]
    st = mc.method_statistics(recs)
    assert st["success_rate"] == 0.75 and st["total_runs"] == 4
    assert np.isclose(st["mean_trans_error"], 0.25) and np.isclose(st["std_trans_error"], np.std([0.1, 0.2, 0.3, 0.4]))  # population std
    assert st["min_rot_error"] == 1.0 and st["max_rot_error"] == 4.0 and st["std_time_ms"] == 0.0
    assert st["mean_iterations"] == 15.0


@pytest.mark.timeout(300)
def test_two_rank_gather_equals_single_process():
    import torch.multiprocessing as tmp
    single_recs, single_stats = mc.run_montecarlo(_oracle_runner(), BASE, N_TRIALS, seed=123, trans_amp=0.3,
                                                  rot_amp_rad=h.deg2rad(1.0))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = tmp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, recs, stats in got:
        assert recs.shape == (N_TRIALS, mc.REC)
        assert np.array_equal(recs[:, mc.R_TRIAL], np.arange(N_TRIALS))
        # identical trial results no matter which rank computed them (deterministic), hence identical statistics
        assert np.array_equal(np.delete(recs, mc.R_TIME, 1), np.delete(single_recs, mc.R_TIME, 1))
        assert stats == single_stats
    assert 0 < single_stats["converged_runs"] <= N_TRIALS


def _worker8(rank, world, port, q):
    """The native driver's data path without a GPU: shard, per-rank records (synthetic: a function of the trial id), gather, statistics."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 45                                            # not a multiple of 8: ragged shards, padded blocks in the gather
    mine = mc.shard_indices(n, rank, world)
    local = np.zeros((len(mine), mc.REC))
    local[:, mc.R_TRIAL] = mine
    local[:, mc.R_ITERS] = 3 + mine % 5
    local[:, mc.R_CONV] = mine % 2
    local[:, mc.R_TERR] = 0.01 * mine
    for j, k in enumerate(mine):
        local[j, mc.R_T:mc.R_T + 16] = mc.trial_pose(BASE, int(k), 99, 0.3, 0.01).reshape(16)      # every rank draws trial k's pose itself
    allr = mc.gather_records(local, n, dist, "cpu")
    q.put((rank, allr, mc.method_statistics(allr)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_world_of_eight_shards_cover_and_gather():
    """8 ranks (the node the scaling curve is for): k = rank mod 8 covers every trial exactly once, the padded fixed-size blocks
    gather into one table ordered by trial on EVERY rank, seeds depend on the trial only."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    q = ctx.Queue()
    world = 8
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cover = np.concatenate([mc.shard_indices(45, r, world) for r in range(world)])
    assert sorted(cover.tolist()) == list(range(45))
    ref = sorted(got)[0][1]
    assert np.array_equal(ref[:, mc.R_TRIAL], np.arange(45)) and np.array_equal(ref[:, mc.R_ITERS], 3 + np.arange(45) % 5)
    for k in (0, 7, 44):
        assert np.array_equal(ref[k, mc.R_T:mc.R_T + 16].reshape(4, 4), mc.trial_pose(BASE, k, 99, 0.3, 0.01))
    for rank, allr, stats in got:
        assert np.array_equal(allr, ref) and stats["total_runs"] == 45 and stats["converged_runs"] == 22


def test_host_share_of_a_rank():
    """hostinfo: a rank's OpenMP team is its share of the CPUs the container may really use (cgroup quota and affinity, not the machine's
    hardware threads), and dcreg_set_host_threads overrides the OMP_NUM_THREADS=1 a launcher exports."""
    from dcreg_amd import api, hostinfo
    n = hostinfo.usable_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    q = hostinfo.cgroup_cpu_quota()
    assert q is None or n <= max(1, int(q))
    assert hostinfo.threads_per_rank(1) == min(32, n) and hostinfo.threads_per_rank(8) == max(1, min(32, n // 8))
    assert hostinfo.threads_per_rank(10 ** 6) == 1
    before = api.load().dcreg_get_host_threads()
    try:
        assert api.set_host_threads(3) == 3 and api.load().dcreg_get_host_threads() == 3
        with pytest.raises(api.DcregError):
            api.set_host_threads(0)
    finally:
        api.set_host_threads(max(before, 1))
