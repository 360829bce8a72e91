#!/usr/bin/env python
"""bench.py -- ICP iterations/s of the MI355X-native DCReg hot path (BASELINE.json metric).

A "step" is ONE ICP iteration of one scan pair: dcreg_linearize (exact 5-NN + plane fit + Jacobian/residual
+ J^T J / J^T r on the GPU, inputs resident in HBM) followed by the host 6x6 Schur analysis + PCG solve and
the SE(3) update -- exactly the reference's per-iteration loop body (icp_test_runner.cpp:1694-2004).
Default workload = BASELINE.json configs[1]: 100k-point synthetic cylinder pair, runs of 20 ICP iterations.
With --gpus N every rank runs its own scan pair (weak scaling, no data-path collective); RCCL is used only
for the final statistics gather.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BYTES_PER_QUERY = 72           # 12 B source point + 5 x 12 B neighbours

WORKLOADS = {
    # name: (scene, n_points, radius, iterations per ICP run)
    "c2_cylinder_100k": ("cylinder", 100_000, 1.0, 20),
    "c4_corridor_1m": ("corridor", 1_000_000, 1.0, 50),
    "c3_planes_200k": ("planes", 200_000, 0.5, 30),
    "c1_fixture_7562": ("fixture", 7562, 1.0, 30),
    # Monte-Carlo: 256 independent trials of the fixture pair advance in lock-step; ONE step = one batched launch
    # = 256 ICP iterations (dcreg_icp_run_trials / dcreg_linearize_batch)
    "c5_montecarlo_fixture": ("fixture", 7562, 1.0, 30),
}
MC_BATCH = 256


def make_pair(scene, n, seed):
    import helpers as h
    if scene == "fixture":
        pts = h.cylinder_cloud()
        return pts, pts.copy()
    gen = {"cylinder": lambda: h.scene_cylinder(n, seed=seed, noise=0.01),
           "corridor": lambda: h.scene_corridor(n, seed=seed),
           "planes": lambda: h.scene_planes(n, seed=seed)}[scene]
    tgt = gen()
    rng = np.random.default_rng(seed + 1000)
    src = (tgt + rng.normal(0, 0.01, tgt.shape)).astype(np.float32)   # a second noisy scan of the same scene
    return tgt, src


KERNEL_TIMING_STRIDE = 8   # HIP-event pair around every 8th launch of the timed region


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--workload", default="c2_cylinder_100k", choices=list(WORKLOADS))
    ap.add_argument("--method", default="Ours")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--sharding", default="pairs", choices=["pairs", "points"],
                    help="pairs (default): one independent scan pair per GPU, no data-path collective, weak scaling; "
                         "points: ONE pair, source points split over the GPUs, one 256 B all_gather per iteration "
                         "(dcreg_amd/pointshard.py), strong scaling")
    ap.add_argument("--concurrent-pairs", type=int, default=4,
                    help="after the main (one pair at a time) measurement, also time P independent scan pairs running "
                         "CONCURRENTLY on this GPU (one context + stream + host thread each) and report the aggregate as "
                         "'concurrent_pairs'; 0 = skip")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE",
                    help="backend option (dcreg_backend_set_option), e.g. --opt warm_start=0 --opt cell_factor=1.5 (ablations)")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hooks (control-flow check of the multi-rank path on a single-GPU box): several ranks on one device over gloo
    backend = os.environ.get("DCREG_BENCH_BACKEND", "nccl")
    if "DCREG_BENCH_LOCAL_RANK" in os.environ:
        local_rank = int(os.environ["DCREG_BENCH_LOCAL_RANK"])
    cdev = "cuda" if backend == "nccl" else "cpu"        # device of the (tiny) collective payloads
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible and there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # RCCL over xGMI
        else:
            dist.init_process_group(backend)
    n_gpus = world

    import helpers as h
    import dcreg_amd
    from dcreg_amd import api

    scene, n_pts, radius, run_len = WORKLOADS[args.workload]
    by_points = args.sharding == "points"
    tgt, src = make_pair(scene, n_pts, seed=100 + (0 if by_points else rank))   # pairs: every rank its own scan pair
    n_src_total = len(src)
    if by_points:                                                # points: the same pair everywhere, this rank's slice
        from dcreg_amd import pointshard
        lo, hi = pointshard.slice_of(len(src), rank, world)
        src = np.ascontiguousarray(src[lo:hi])
        reducer = pointshard.make_reducer(dist, cdev)
    ctx = dcreg_amd.Context(local_rank)
    for kv in args.opt:
        k, v = kv.split("=", 1)
        ctx.set_option(k, float(v))
    ctx.set_target(tgt, radius)
    ctx.set_source(src)
    info = ctx.index_info()
    det, hand = api.METHODS[args.method]
    cfg = api.default_config(search_radius=radius, max_iterations=run_len, KAPPA_TARGET=10.0, STD_REG_GAMMA=100.0,
                             CONVERGENCE_THRESH_ROT=0.0, CONVERGENCE_THRESH_TRANS=0.0,   # fixed-length runs
                             use_weight_derivative=1, always_compute_schur=1)
    prm = api.default_lin_params(radius, 1)
    # initial misalignment of every run: a few cm / tenths of a degree (frame-to-frame LiDAR odometry regime);
    # the synthetic pair converges from it, so every iteration keeps ~all correspondences alive
    T_init = h.pose6d_matrix(0.05, -0.08, 0.03, h.deg2rad(0.2), h.deg2rad(-0.1), h.deg2rad(0.5))
    import ctypes as C
    L = api.load()
    dp = C.POINTER(C.c_double)
    R0 = np.ascontiguousarray(T_init[:3, :3]).reshape(9).copy()
    t0 = T_init[:3, 3].copy()
    res = api.IcpResult()
    state = {"done": 0, "mc_iters": 0}

    mc = args.workload.startswith("c5_")
    if mc:
        from dcreg_amd import montecarlo as mcm
        base = (0.2, 0.8, 0.5, h.deg2rad(0.1), h.deg2rad(0.1), h.deg2rad(2.0))
        T0s = np.stack([mcm.trial_pose(base, k + 1000 * rank, 2024, 0.3, h.deg2rad(1.0)) for k in range(MC_BATCH)])
        R0s = np.ascontiguousarray(T0s[:, :3, :3]).reshape(MC_BATCH, 9)
        t0s = np.ascontiguousarray(T0s[:, :3, 3]).reshape(MC_BATCH, 3)
        trial_res = (api.TrialResult * MC_BATCH)()

    def run_steps_mc(k):
        """k lock-step iterations of MC_BATCH independent trials (each iteration = one batched launch)."""
        left = k
        while left > 0:
            n = min(run_len, left)
            cfg.max_iterations = n
            rc = L.dcreg_icp_run_trials(ctx._h, MC_BATCH, R0s.ctypes.data_as(dp), t0s.ctypes.data_as(dp), api.DETECTION[det],
                                        api.HANDLING[hand], C.byref(cfg), trial_res)
            if rc != 0:
                raise RuntimeError("dcreg_icp_run_trials failed: %s" % L.dcreg_last_error(ctx._h))
            state["mc_iters"] += sum(trial_res[i].iterations for i in range(MC_BATCH))   # trials that abort stop counting
            left -= n

    def run_steps_points(k):
        """k lock-step iterations of ONE scan pair whose source points are split over the ranks: per iteration one
        linearisation of the local slice, one all_gather of 32 doubles, the host step on every rank."""
        left = k
        while left > 0:
            n = min(run_len, left)
            cfg.max_iterations = n
            out, _ = ctx.icp_run_sharded(T_init, args.method, cfg, n_src_total, reducer, log_capacity=0)
            if out.iterations != n or out.status != 0:
                raise RuntimeError("point-sharded run stopped early: iterations=%d status=%d" % (out.iterations, out.status))
            res.R[:] = out.R[:]; res.t[:] = out.t[:]
            left -= n
        state["done"] += k

    def run_steps(k):
        if mc:
            return run_steps_mc(k)
        if by_points:
            return run_steps_points(k)
        return run_steps_single(k)

    def run_steps_single(k):
        """k ICP iterations through the product's engine seam (dcreg_icp_run: device linearisation + host
        Schur analysis / PCG / SE(3) update per iteration, all in C++), as runs of `run_len` iterations from the
        initial pose; convergence thresholds are 0 so every run has exactly its max_iterations iterations."""
        left = k
        while left > 0:
            n = min(run_len, left)
            cfg.max_iterations = n
            rc = L.dcreg_icp_run(ctx._h, R0.ctypes.data_as(dp), t0.ctypes.data_as(dp), api.DETECTION[det], api.HANDLING[hand],
                                 C.byref(cfg), None, 0, C.byref(res))
            if rc != 0 or res.iterations != n:
                raise RuntimeError("dcreg_icp_run failed: rc=%d iterations=%d status=%d %s" % (rc, res.iterations, res.status, L.dcreg_last_error(ctx._h)))
            left -= n
        state["done"] += k

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(args.warmup)
    # HIP events around every 8th linearisation of the timed region (an event pair costs ~10 us of host time per
    # launch: bracketing every launch would slow the measured loop by ~25 %)
    ctx.set_option("time_kernels", KERNEL_TIMING_STRIDE)
    ctx.kernel_time(reset=True)
    fence()
    mc_before = state["mc_iters"]
    t_start = time.perf_counter()
    run_steps(args.steps)
    fence()
    elapsed = time.perf_counter() - t_start
    kern_ms, kern_n = ctx.kernel_time(reset=True)
    ctx.set_option("time_kernels", 0)
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- throughput with several independent pairs in flight on the same GPU (extra figure, not `value`): at 100 k points
    # one linearisation occupies 1.5 of the 4 waves/SIMD the register budget allows and the device idles during every
    # host step, so independent pairs interleave almost for free
    conc = None
    P = args.concurrent_pairs
    if P > 1 and not mc and not by_points and n_gpus == 1:     # (single-GPU runs only, like the CPU baseline: it needs P host threads)
        import threading
        ctxs = [ctx]
        for q in range(1, P):
            tq, sq = make_pair(scene, n_pts, seed=100 + rank + 1000 * q)
            cq = dcreg_amd.Context(local_rank)
            for kv in args.opt:
                k2, v2 = kv.split("=", 1)
                cq.set_option(k2, float(v2))
            cq.set_target(tq, radius); cq.set_source(sq)
            ctxs.append(cq)

        def pair_worker(cq, k, barrier):
            cfg_q = api.default_config(search_radius=radius, max_iterations=run_len, KAPPA_TARGET=10.0, STD_REG_GAMMA=100.0,
                                       CONVERGENCE_THRESH_ROT=0.0, CONVERGENCE_THRESH_TRANS=0.0, use_weight_derivative=1, always_compute_schur=1)
            res_q = api.IcpResult()
            barrier.wait()
            left = k
            while left > 0:
                n = min(run_len, left)
                cfg_q.max_iterations = n
                rc = L.dcreg_icp_run(cq._h, R0.ctypes.data_as(dp), t0.ctypes.data_as(dp), api.DETECTION[det], api.HANDLING[hand],
                                     C.byref(cfg_q), None, 0, C.byref(res_q))
                if rc != 0 or res_q.iterations != n:
                    pair_errors.append("concurrent pair failed: rc=%d iterations=%d %s" % (rc, res_q.iterations, L.dcreg_last_error(cq._h)))
                    return
                left -= n

        pair_errors = []
        t_conc = None
        for k in (args.warmup, args.steps):
            barrier = threading.Barrier(P + 1)
            th = [threading.Thread(target=pair_worker, args=(cq, k, barrier)) for cq in ctxs]
            for t_ in th:
                t_.start()
            fence()
            barrier.wait()
            ta = time.perf_counter()
            for t_ in th:
                t_.join()
            fence()
            t_conc = time.perf_counter() - ta
        if pair_errors:
            raise RuntimeError("; ".join(pair_errors))
        if dist is not None:
            tm = torch.tensor([t_conc], dtype=torch.float64, device=cdev)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            t_conc = float(tm.item())
        conc = {"pairs_per_gpu": P, "value": n_gpus * P * args.steps / t_conc, "unit": "iterations/s",
                "us_per_iteration_per_pair": 1e6 * t_conc / args.steps,
                # device-level: algorithmic bytes of all P pairs' launches over the WALL time of the region (host steps included)
                "achieved_GBps_wall": P * BYTES_PER_QUERY * n_pts * args.steps / t_conc / 1e9,
                "frac_of_hbm_peak_wall": P * BYTES_PER_QUERY * n_pts * args.steps / t_conc / 1e9 / HBM_PEAK_GBS,
                "note": "P independent %d-pt pairs in flight per GPU, one context + stream + host thread each; aggregate over all GPUs" % n_pts}
        for cq in ctxs[1:]:
            cq.close()

    # final statistics gather (the only collective): per-rank pose error / rmse / correspondences
    T_fin = np.eye(4); T_fin[:3, :3] = np.array(res.R[:]).reshape(3, 3); T_fin[:3, 3] = res.t[:]
    te, re_ = api.pose_error(np.eye(4), T_fin)
    last = ctx.linearize(T_fin[:3, :3], T_fin[:3, 3], prm)
    rec = torch.tensor([te, re_, float(last["n_eff"]), kern_ms / max(kern_n, 1)], dtype=torch.float64, device=cdev)
    if dist is not None:
        allrec = [torch.zeros_like(rec) for _ in range(world)]
        dist.all_gather(allrec, rec)
        recs = torch.stack(allrec).cpu().numpy()
    else:
        recs = rec.cpu().numpy()[None, :]

    if rank == 0:
        per_step = (state["mc_iters"] - mc_before) / args.steps if mc else 1
        iters_per_s = (1 if by_points else n_gpus) * args.steps * per_step / elapsed
        kern_us = float(np.mean(recs[:, 3])) * 1e3
        algo_bytes = BYTES_PER_QUERY * len(src) * per_step          # per launch of THIS kernel (a rank's slice when sharded by points)
        traffic = measured_traffic(args.workload)
        achieved = algo_bytes / (kern_us * 1e-6) / 1e9 if kern_us > 0 else 0.0
        result = {
            "metric": "ICP iterations/sec", "value": iters_per_s, "unit": "iterations/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong" if by_points else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: %d-pt source x %d-pt target, radius %.2f, runs of %d ICP iterations, method %s "
                                   "(Schur detection + PCG), %s" % (args.workload, n_src_total, len(tgt), radius, run_len, args.method,
                                                                    "ONE scan pair, source points split over the GPUs" if by_points else "one scan pair per GPU"),
                       "n_src": int(n_src_total), "n_tgt": int(len(tgt)), "grid_cell_m": info.cell, "grid_cells": int(info.n_cells)},
            "correspondence_queries_per_s": iters_per_s * n_src_total, "icp_iterations_per_step": per_step,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "k_linearize (fused exact 5-NN + plane fit + point-to-plane row + J^T J / J^T r reduction)",
                         "kernel_us_avg": kern_us, "algorithmic_bytes_per_launch": algo_bytes},
            "final_stats": {"mean_trans_error_m": float(np.mean(recs[:, 0])), "mean_rot_error_deg": float(np.mean(recs[:, 1])),
                            "mean_correspondences": float(np.mean(recs[:, 2]))},
        }
        if conc is not None:
            result["concurrent_pairs"] = conc
        if n_gpus == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(tgt, src, T_init, radius, run_len, args.method, args.cpu_seconds)
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


def measured_traffic(workload):
    """HBM-side bytes per k_linearize launch from the rocprofv3 PMC passes of this workload (FETCH_SIZE / WRITE_SIZE,
    separate passes, KB -> bytes, read side x2 per the gfx950 note in MI355X_MICROARCH.md), recorded by
    scripts/collect_profiles.sh + scripts/summarize_profiles.py under profiles/.  None if no profile is committed."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s.json" % workload))):
        try:
            t = json.load(open(f)).get("traffic")
            if t:
                best = t["bytes_fetch_x2"]
        except Exception:
            pass
    return best


def cpu_baseline(tgt, src, T_init, radius, run_len, method, budget_s):
    """The CPU oracle (oracle/, a C/OpenMP restatement of the reference path; the reference itself needs
    Eigen/PCL/FLANN and cannot be built here) timed on this box's host cores on a bounded sample: first the reference's own
    configuration (ONE pair, the 8 OpenMP threads it hard-codes, :1714), then the whole box as cores/8 concurrent
    8-thread runs of the same pair (the CPU counterpart of the GPU's concurrent_pairs figure)."""
    import threading
    from oracle import pyoracle as po
    tree = po.KdTree(tgt)                       # kd-tree build is untimed in the reference too (:408-442)
    ncpu = os.cpu_count() or 1
    threads = min(8, ncpu)

    def run(budget, counter, slot, barrier=None):
        cfg = po.default_config(search_radius=radius, max_iterations=1, thresh_rot=0.0, thresh_trans=0.0, kappa_target=10.0,
                                std_reg_gamma=100.0, use_weight_derivative=1, always_compute_schur=1, num_threads=threads)
        T = T_init.copy()
        po.icp_run(tree, src, T, method, cfg)       # warm-up (thread pool, page faults)
        if barrier is not None:
            barrier.wait()
        n = 0
        t0 = time.perf_counter()
        while True:
            res, logs = po.icp_run(tree, src, T, method, cfg)
            T = np.eye(4); T[:3, :3] = np.array(res.R[:]).reshape(3, 3); T[:3, 3] = res.t[:]
            n += 1
            if n % run_len == 0:
                T = T_init.copy()
            el = time.perf_counter() - t0
            if el > budget or n >= 100 * run_len:
                break
        counter[slot] = (n, el)

    one = [None]
    run(budget_s / 2, one, 0)
    n, el = one[0]
    out = {"value": n / el, "unit": "iterations/s", "cores": threads, "kind": "port",
           "sample": "%d ICP iterations of the same scan pair (%d-pt source), one pair at a time, OpenMP x%d, %.1f s" % (n, len(src), threads, el)}
    teams = max(1, ncpu // threads)
    if teams > 1:
        res = [None] * teams
        barrier = threading.Barrier(teams)
        th = [threading.Thread(target=run, args=(budget_s / 2, res, i, barrier)) for i in range(teams)]
        c0, w0 = os.times(), time.perf_counter()
        for t_ in th:
            t_.start()
        for t_ in th:
            t_.join()
        c1, w1 = os.times(), time.perf_counter()
        busy = ((c1.user + c1.system) - (c0.user + c0.system)) / max(w1 - w0, 1e-9)     # CPUs this process actually got
        agg = sum(r[0] / r[1] for r in res if r)
        out["all_cores"] = {"value": agg, "unit": "iterations/s", "cores": teams * threads, "cpus_obtained": busy,
                            "sample": "%d concurrent runs of the same pair x %d OpenMP threads each" % (teams, threads)}
        out["sample"] += "; %d such runs side by side on %d hardware threads (%.0f CPUs obtained): %.1f it/s aggregate" % (
            teams, teams * threads, busy, agg)
    return out


if __name__ == "__main__":
    main()
