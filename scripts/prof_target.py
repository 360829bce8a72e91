"""Small fixed workload for rocprofv3 runs: N linearisations of one synthetic pair (GPU box only)."""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dcreg_amd import scenes as h
import dcreg_amd
from dcreg_amd import api

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="corridor")
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--tile", type=int, default=1)
ap.add_argument("--cell-factor", type=float, default=0.0)
args = ap.parse_args()
gen = {"corridor": h.scene_corridor, "cylinder": lambda n, seed: h.scene_cylinder(n, seed=seed, noise=0.01), "planes": h.scene_planes}[args.scene]
tgt = gen(args.n, seed=1)
rng = np.random.default_rng(0)
src = (tgt + rng.normal(0, 0.01, tgt.shape)).astype(np.float32)
ctx = dcreg_amd.Context(0)
if args.cell_factor > 0:
    ctx.set_option("cell_factor", args.cell_factor)
ctx.set_target(tgt, 1.0)
ctx.set_source(src)
T0 = h.pose6d_matrix(0.05, -0.08, 0.03, 0.003, -0.002, 0.008)
R = np.ascontiguousarray(T0[:3, :3]).reshape(9); t = T0[:3, 3].copy()
prm = api.default_lin_params(1.0, 1); out = api.LinOut()
ctx.set_option("time_kernels", 1)
for _ in range(args.iters):
    ctx.linearize_raw(R, t, prm, out)
ms, n = ctx.kernel_time()
info = ctx.index_info()
print("n_eff", out.n_eff, "kernel us", ms / n * 1e3, "cell", info.cell, "cells", info.n_cells)
