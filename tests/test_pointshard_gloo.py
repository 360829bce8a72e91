"""Point sharding of one scan pair (dcreg_amd/pointshard.py): world_size-2 gloo run on CPU.  The per-slice
linearisation is played by the oracle here (tests may use it); on the GPU node it is Context.linearize of a context
whose source is the rank's slice, and the 32-double rows travel over RCCL.  The host step uses the product's solver
seam (pure host functions of the C-ABI library: no GPU needed)."""
import os
import socket

import numpy as np
import pytest

import helpers as h
from dcreg_amd import api, pointshard as ps

T_INIT = h.pose6d_matrix(**h.PAPER_INIT)


def _cfg():
    return api.default_config(search_radius=1.0, max_iterations=12, CONVERGENCE_THRESH_TRANS=1e-3, CONVERGENCE_THRESH_ROT=1e-5,
                              KAPPA_TARGET=10.0, STD_REG_GAMMA=100.0, use_weight_derivative=1, always_compute_schur=1)


def _oracle_linearize(lo, hi):
    from oracle import pyoracle as po
    pts = h.cylinder_cloud()
    tree = po.KdTree(pts)
    src = np.ascontiguousarray(pts[lo:hi])
    prm = po.default_lin_params(1.0, 1, num_threads=2)
    return lambda R, t: po.linearize(tree, src, R, t, prm)


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = len(h.cylinder_cloud())
    lo, hi = ps.slice_of(n, rank, world)
    res = ps.icp_run(_oracle_linearize(lo, hi), n, T_INIT, "Ours", _cfg(), dist=dist)
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_slices_cover_the_cloud():
    for n, w in ((7562, 2), (10, 3), (5, 8), (1_000_000, 8)):
        sl = [ps.slice_of(n, r, w) for r in range(w)]
        assert sl[0][0] == 0 and sl[-1][1] == n and all(a[1] == b[0] for a, b in zip(sl, sl[1:]))
        assert max(hi - lo for lo, hi in sl) - min(hi - lo for lo, hi in sl) <= 1


def test_sum_of_slice_linearisations_is_the_whole():
    from oracle import pyoracle as po
    pts = h.cylinder_cloud()
    whole = ps.pack_row(_oracle_linearize(0, len(pts))(T_INIT[:3, :3], T_INIT[:3, 3]))
    parts = sum(ps.pack_row(_oracle_linearize(*ps.slice_of(len(pts), r, 3))(T_INIT[:3, :3], T_INIT[:3, 3])) for r in range(3))
    assert whole[29] == parts[29] == 197 and whole[30] == parts[30]            # N_eff of the paper trace, iteration 0
    assert h.rel_err(parts[:21], whole[:21]) < 1e-12 and h.rel_err(parts[21:27], whole[21:27]) < 1e-11


@pytest.mark.timeout(300)
def test_two_rank_point_sharded_run_equals_single_process_and_the_committed_trace():
    import torch.multiprocessing as tmp
    n = len(h.cylinder_cloud())
    single = ps.icp_run(_oracle_linearize(0, n), n, T_INIT, "Ours", _cfg())
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
    a, b = got[0][1], got[1][1]
    # both ranks hold the identical state (no broadcast needed): bitwise equal poses and logs
    assert np.array_equal(a["T"], b["T"]) and a["iterations"] == b["iterations"] == single["iterations"]
    assert a["converged"] == single["converged"] and a["status"] == 0
    for la, lb, ls in zip(a["log"], b["log"], single["log"]):
        assert np.array_equal(la["dx"], lb["dx"]) and la["n_eff"] == lb["n_eff"] == ls["n_eff"]
        assert np.allclose(la["dx"], ls["dx"], rtol=0, atol=1e-9)          # association order of the fp64 sums only
    # the committed "Ours" trace of the paper run (10 iterations): N_eff exact, update to the print precision
    gold = h.golden_rows("paper", "iteration_details_with_dx.csv", "Ours")
    cond = h.golden_rows("paper", "condition_numbers_detailed.csv", "Ours")
    assert a["iterations"] == len(gold)
    for L, g, c in zip(a["log"], gold, cond):
        assert L["n_eff"] == int(c["Effective_Points"])
        assert np.allclose(L["dx"], [float(g[k]) for k in ("dx_wx", "dx_wy", "dx_wz", "dx_x", "dx_y", "dx_z")], rtol=0, atol=5e-7)
