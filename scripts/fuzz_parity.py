"""Randomised parity hunt (GPU box): random scenes, sizes, radii, cell sizes and pose sequences (so that cold, warm, ring-walk
and empty-space paths all occur); neighbour indices, float distances and gate flags must equal the oracle's bit for bit,
H / g to 1e-8.  usage: fuzz_parity.py [n_cases] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dcreg_amd import scenes as h


def _rel_err(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))

import dcreg_amd
from dcreg_amd import api
from oracle import pyoracle as po

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = dcreg_amd.Context(0)
bad = 0
for case in range(n_cases):
    kind = rng.integers(0, 4)
    n = int(rng.choice([800, 3000, 12000, 40000]))
    if kind == 0: tgt = h.scene_cylinder(n, seed=int(rng.integers(1 << 30)), noise=float(rng.choice([0.0, 0.01, 0.05])))
    elif kind == 1: tgt = h.scene_corridor(n, seed=int(rng.integers(1 << 30)), length=float(rng.choice([20.0, 60.0])))
    elif kind == 2: tgt = h.scene_planes(n, seed=int(rng.integers(1 << 30)))
    else: tgt = (rng.uniform(-3, 3, (n, 3)) * np.array([1.0, 1.0, float(rng.choice([0.02, 1.0]))])).astype(np.float32)
    m = int(rng.integers(200, 3000))
    src = tgt[rng.integers(0, len(tgt), m)] + rng.normal(0, float(rng.choice([0.0, 0.01, 0.2])), (m, 3))
    if rng.random() < 0.3:
        src = np.concatenate([src, rng.uniform(-60, 60, (50, 3))])          # far outliers
    src = src.astype(np.float32)
    radius = float(rng.choice([0.3, 0.5, 1.0, 2.0]))
    ctx.set_option("cell_factor", float(rng.choice([1.0, 1.5, 2.0, 3.0])))
    ctx.set_option("gap_field", int(rng.integers(0, 2)))
    ctx.set_option("warm_start", int(rng.integers(0, 2)))
    ctx.set_target(tgt, radius); ctx.set_source(src)
    tree = po.KdTree(tgt)
    wd = int(rng.integers(0, 2))
    for step in range(4):
        amp = float(rng.choice([0.005, 0.05, 0.5]))
        T = h.pose6d_matrix(*(rng.normal(0, amp, 3)), *(rng.normal(0, amp * 0.05, 3)))
        g = ctx.linearize(T[:3, :3], T[:3, 3], api.default_lin_params(radius, wd), debug=True)
        r = po.linearize(tree, src, T[:3, :3], T[:3, 3], po.default_lin_params(radius, wd), debug=True)
        ok = r["flag"] != 0
        good = (np.array_equal(g["flag"], r["flag"]) and np.array_equal(g["nn_idx"][ok], r["nn_idx"][ok]) and
                np.array_equal(g["nn_d2"][ok].view(np.uint32), r["nn_d2"][ok].view(np.uint32)) and g["n_eff"] == r["n_eff"] and g["n_pt"] == r["n_pt"])
        if good and r["n_eff"] > 0:
            good = _rel_err(g["H_upper"], r["H_upper"]) < 1e-8 and _rel_err(g["g"], r["g"]) < 1e-7
        if not good:
            bad += 1
            print("MISMATCH case %d step %d: kind %d n %d m %d radius %.2f n_eff %d/%d flags equal %s" % (
                case, step, kind, len(tgt), len(src), radius, g["n_eff"], r["n_eff"], np.array_equal(g["flag"], r["flag"])), flush=True)
    print("case %2d ok: kind %d, %5d x %4d, R %.1f, cell %.3f, n_eff %d" % (case, kind, len(tgt), len(src), radius, ctx.index_info().cell, r["n_eff"]), flush=True)
print("fuzz done: %d mismatches in %d cases" % (bad, n_cases))
sys.exit(1 if bad else 0)
