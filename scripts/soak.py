"""Soak test (GPU box): many context lifetimes with clouds of changing size, mixed entry points; device memory must return to
its starting level and nothing may hang."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dcreg_amd import scenes as h
import dcreg_amd
from dcreg_amd import api

rng = np.random.default_rng(0)
free0 = torch.cuda.mem_get_info()[0]
t0 = time.perf_counter()
cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 150
iters = 0
for c in range(cycles):
    ctx = dcreg_amd.Context(0)
    for _kv in os.environ.get("DCREG_FUZZ_OPTS", "").split():      # e.g. DCREG_FUZZ_OPTS="advance=2 team_pass=0": the hunt with a pass forced
        ctx.set_option(_kv.split("=")[0], float(_kv.split("=")[1]))
    for rep in range(int(rng.integers(1, 4))):
        n = int(rng.choice([500, 7000, 60000, 250000]))
        tgt = h.scene_cylinder(n, seed=int(rng.integers(1 << 30)), noise=0.01)
        src = tgt[rng.permutation(n)[: max(n // 3, 100)]].copy()
        ctx.set_target(tgt, float(rng.choice([0.5, 1.0])))
        ctx.set_source(src)
        T0 = h.pose6d_matrix(0.02, -0.03, 0.01, 0.001, -0.001, 0.003)
        cfg = api.default_config(search_radius=1.0, max_iterations=int(rng.integers(3, 15)), KAPPA_TARGET=10.0, STD_REG_GAMMA=100.0,
                                 use_weight_derivative=int(rng.integers(0, 2)), always_compute_schur=1)
        mode = rng.integers(0, 4)
        if mode == 0:
            res, _ = ctx.icp_run(T0, "Ours", cfg); iters += res.iterations
        elif mode == 1:
            rs = ctx.icp_run_trials(np.stack([T0] * int(rng.choice([3, 70, 200]))), "ME-SR", cfg); iters += sum(r.iterations for r in rs)
        elif mode == 2:
            res, _, _ = ctx.icp_run_euler((0.001, -0.001, 0.003, 0.02, -0.03, 0.01), "ME-TSVD", cfg); iters += res.iterations
        else:
            ctx.linearize(T0[:3, :3], T0[:3, 3], debug=True); ctx.p2p_error(T0, 0.2); ctx.knn(src[:100], k=5, max_radius=0.0); iters += 1
    ctx.close()
    if c == 0:
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]      # baseline after the runtime's one-time allocations (code objects, pools)
    if c % 25 == 0:
        torch.cuda.synchronize()
        print("cycle %d: free memory delta %.1f MB, %d ICP iterations so far, %.1f s" % (c, (free0 - torch.cuda.mem_get_info()[0]) / 1e6, iters, time.perf_counter() - t0), flush=True)
torch.cuda.synchronize()
leak = (free0 - torch.cuda.mem_get_info()[0]) / 1e6
print("soak done: %d cycles, %d ICP iterations, %.1f s, device memory delta %.1f MB" % (cycles, iters, time.perf_counter() - t0, leak))
sys.exit(1 if leak > 64 else 0)
