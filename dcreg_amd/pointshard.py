"""Point sharding of ONE scan pair over several GPUs (second way; config C4 at > 1 GPU).

Every rank holds the whole target (its own index) and a slice of the source points.  One ICP iteration is

    rank r :  dcreg_linearize on its slice                    -> 31 sums (21 H + 6 g + sum r^2 + sum b^2 + n_eff + n_pt)
    all    :  ONE all_gather of 32 doubles per rank           (the path's only exchange step; RCCL over xGMI on the GPU
                                                               node, gloo in the CPU tests; 256 B -> latency-bound)
    all    :  add the rank rows in rank order                 -> identical totals on every rank, independent of timing
    all    :  host analyse / solve / SE(3) update (solver seam of the C-ABI) -> identical next pose on every rank

so the ranks advance in lock-step without a broadcast.  Two drivers: `icp_run` below (Python loop over any `linearize`
callable; what the CPU tests exercise), `Context.icp_run_sharded` + `make_reducer` (the loop runs in the C++ engine,
`dcreg_icp_run_sharded`, and calls back once per iteration for the exchange: any torch.distributed backend) and
`init_native_exchange` + `Context.icp_run_sharded_rccl` (the exchange is an ncclAllGather issued by the engine itself on the
ctx's stream: what bench.py --sharding points uses on the GPU node).  Semantics per iteration are those of dcreg_icp_run
(DCReg/src/icp_test_runner.cpp:1611-2060): n_eff < 10 abort, non-finite update abort, convergence on |d omega|, |d t|.
The sum of slice linearisations equals the linearisation of the whole cloud up to the association order of the fp64
sums (tested: 1e-12 relative).  Trial / scan-pair sharding (dcreg_amd/montecarlo.py, bench.py --gpus N) remains the
primary multi-GPU mode: it needs no per-iteration exchange at all.
"""
import numpy as np

from . import api

ROW = 32   # doubles per rank per iteration


def slice_of(n_points, rank, world):
    """Contiguous slice [lo, hi) of the source points owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(int(n_points), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_row(out):
    """dict / LinOut with H_upper, g, sum_r2, sum_b2, n_eff, n_pt -> float64[32]"""
    get = (lambda k: out[k]) if isinstance(out, dict) else (lambda k: getattr(out, k))
    r = np.zeros(ROW)
    r[:21] = np.asarray(get("H_upper"), np.float64).reshape(21)
    r[21:27] = np.asarray(get("g"), np.float64).reshape(6)
    r[27], r[28], r[29], r[30] = get("sum_r2"), get("sum_b2"), get("n_eff"), get("n_pt")
    return r


def reduce_rows(local_row, dist=None, device="cpu"):
    """all_gather the 32-double rows and add them in rank order (deterministic, identical on every rank)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return np.array(local_row, np.float64)
    import torch
    world = dist.get_world_size()
    mine = torch.as_tensor(np.asarray(local_row, np.float64), device=device)
    rows = torch.empty(world * ROW, dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(rows, mine)
    rows = rows.cpu().numpy().reshape(world, ROW)
    total = np.zeros(ROW)
    for r in range(world):          # fixed association order
        total = total + rows[r]
    return total


def make_reducer(dist=None, device="cpu"):
    """In-place reducer for Context.icp_run_sharded (the C++ engine loop calls it once per iteration): all_gather of the
    32-double row on `device`, rows added in rank order.  Reuses its tensors: no allocation in the loop."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return lambda row: None
    import torch
    world = dist.get_world_size()
    mine = torch.empty(ROW, dtype=torch.float64, device=device)
    rows = torch.empty(world * ROW, dtype=torch.float64, device=device)
    stage = torch.empty(ROW, dtype=torch.float64).pin_memory() if str(device).startswith("cuda") else torch.empty(ROW, dtype=torch.float64)

    def reduce_in_place(row):
        stage.numpy()[:] = row
        mine.copy_(stage, non_blocking=False)
        dist.all_gather_into_tensor(rows, mine)
        allr = rows.cpu().numpy().reshape(world, ROW)
        total = np.zeros(ROW)
        for r in range(world):      # fixed association order
            total = total + allr[r]
        row[:] = total
    return reduce_in_place


def init_native_exchange(ctx, dist=None, device="cuda"):
    """Set up the ctx's own RCCL communicator for dcreg_icp_run_sharded_rccl (the exchange then runs inside the C++ engine
    loop, no Python callback per iteration).  Collective: rank 0 draws the 128-byte communicator id, torch.distributed carries
    it to the other ranks (any backend), every rank calls dcreg_comm_init."""
    import torch
    world = dist.get_world_size() if (dist is not None and dist.is_initialized()) else 1
    rank = dist.get_rank() if world > 1 else 0
    buf = torch.zeros(128, dtype=torch.uint8, device=device if world > 1 else "cpu")
    if rank == 0:
        buf.copy_(torch.frombuffer(bytearray(api.comm_unique_id()), dtype=torch.uint8))
    if world > 1:
        dist.broadcast(buf, src=0)
    ctx.comm_init(bytes(buf.cpu().numpy().tobytes()), rank, world)
    return rank, world


def icp_run(linearize, n_src_total, T0, method, cfg, dist=None, device="cpu"):
    """Lock-step ICP over point shards.  `linearize(R, t) -> dict/LinOut` evaluates THIS rank's slice (Context.linearize of a
    context whose source is the slice); n_src_total = points of the whole source cloud.  Returns a dict with the final
    transform, iterations, converged, status and the per-iteration log (identical on every rank)."""
    T0 = np.asarray(T0, np.float64).reshape(4, 4)
    R, t = T0[:3, :3].copy(), T0[:3, 3].copy()
    det, hand = api.METHODS[method] if isinstance(method, str) else method
    res = {"converged": 0, "iterations": 0, "status": 0, "log": []}
    for it in range(cfg.max_iterations):
        tot = reduce_rows(pack_row(linearize(R, t)), dist, device)
        n_eff, n_pt = int(round(tot[29])), int(round(tot[30]))
        if n_eff < 10:                                                   # :1847-1854
            res.update(iterations=it + 1, status=1)
            break
        H = api.unpack_hessian(tot[:21])
        an = api.analyze_degeneracy(H, det, hand, cfg)                   # :1922-1923
        dx = api.solve_degenerate_system(H, tot[21:27], hand, cfg, an)   # :1940
        if not np.all(np.isfinite(dx)):                                  # :1942-1950
            res.update(iterations=it, status=2)
            break
        R, t = api.boxplus(R, t, dx)                                     # :1953
        res["log"].append({"iter": it, "n_eff": n_eff, "n_pt": n_pt, "fitness": n_pt / float(n_src_total),
                           "rmse": float(np.sqrt(tot[27] / n_eff)), "objective": 0.5 * tot[28], "dx": np.array(dx),
                           "mask": [int(m) for m in an.degenerate_mask]})
        res["iterations"] = it + 1
        if np.linalg.norm(dx[:3]) < cfg.CONVERGENCE_THRESH_ROT and np.linalg.norm(dx[3:]) < cfg.CONVERGENCE_THRESH_TRANS:   # :1998
            res["converged"] = 1
            break
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, t
    res["T"] = T
    return res
