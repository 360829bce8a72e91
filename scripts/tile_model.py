"""Host-side model of the per-wave tile boxes (no GPU): distribution of rows / table entries / points."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dcreg_amd import scenes as h

def spread21(v):
    v = v.astype(np.uint64) & np.uint64(0x1FFFFF)
    v = (v | (v << np.uint64(32))) & np.uint64(0x1F00000000FFFF)
    v = (v | (v << np.uint64(16))) & np.uint64(0x1F0000FF0000FF)
    v = (v | (v << np.uint64(8))) & np.uint64(0x100F00F00F00F00F)
    v = (v | (v << np.uint64(4))) & np.uint64(0x10C30C30C30C30C3)
    v = (v | (v << np.uint64(2))) & np.uint64(0x1249249249249249)
    return v

def hilbert_key(q, b=21):
    X = [q[:, 0].astype(np.uint64).copy(), q[:, 1].astype(np.uint64).copy(), q[:, 2].astype(np.uint64).copy()]
    M = np.uint64(1 << (b - 1))
    Q = int(M)
    while Q > 1:
        P = np.uint64(Q - 1); Qm = np.uint64(Q)
        for i in range(3):
            hit = (X[i] & Qm) != 0
            X[0] = np.where(hit, X[0] ^ P, X[0])
            t = np.where(hit, np.uint64(0), (X[0] ^ X[i]) & P)
            X[0] ^= t; X[i] ^= t
        Q >>= 1
    for i in range(1, 3): X[i] ^= X[i - 1]
    t = np.zeros_like(X[0]); Q = int(M)
    while Q > 1:
        t = np.where((X[2] & np.uint64(Q)) != 0, t ^ np.uint64(Q - 1), t)
        Q >>= 1
    for i in range(3): X[i] ^= t
    return (spread21(X[0]) << np.uint64(2)) | (spread21(X[1]) << np.uint64(1)) | spread21(X[2])


def model(tgt, src, hcell, group=64, curve="morton"):
    mn = tgt.min(0).astype(np.float64)
    dims = np.floor((tgt.max(0) - mn) / hcell).astype(int) + 1
    tc = np.clip(np.floor((tgt - mn) / hcell).astype(int), 0, dims - 1)
    lin = (tc[:, 2] * dims[1] + tc[:, 1]) * dims[0] + tc[:, 0]
    counts = np.bincount(lin, minlength=int(np.prod(dims)))
    cum = np.concatenate([[0], np.cumsum(counts)])
    smn = src.min(0); ext = (src.max(0) - smn).max()
    q = np.clip(((src - smn) * (2097151.0 / ext * 0.999999)), 0, 2097151).astype(np.uint64)
    key = spread21(q[:, 0]) | (spread21(q[:, 1]) << np.uint64(1)) | (spread21(q[:, 2]) << np.uint64(2))
    if curve == "hilbert":
        key = hilbert_key(q)
    order = np.argsort(key, kind="stable")
    sc = np.floor((src[order] - mn) / hcell).astype(int)
    n = len(src) // group * group
    sc = sc[:n].reshape(-1, group, 3)
    lo = np.clip(sc.min(1) - 1, 0, dims - 1); hi = np.clip(sc.max(1) + 1, 0, dims - 1)
    W = hi[:, 0] - lo[:, 0] + 1; nyb = hi[:, 1] - lo[:, 1] + 1; nzb = hi[:, 2] - lo[:, 2] + 1
    rows = nyb * nzb; tabn = rows * (W + 1)
    # points in box (sample up to 3000 groups)
    sel = np.random.default_rng(0).choice(len(rows), min(3000, len(rows)), replace=False)
    pts = []
    for gidx in sel:
        tot = 0
        for z in range(lo[gidx, 2], hi[gidx, 2] + 1):
            base = (z * dims[1] + np.arange(lo[gidx, 1], hi[gidx, 1] + 1)) * dims[0]
            tot += int((cum[base + hi[gidx, 0] + 1] - cum[base + lo[gidx, 0]]).sum())
        pts.append(tot)
    pts = np.array(pts)
    return rows, tabn, pts, rows[sel], tabn[sel]

for name, gen, hc in (("cyl100k", lambda: h.scene_cylinder(100_000, seed=1, noise=0.01), 0.84), ("corr1M", lambda: h.scene_corridor(1_000_000, seed=1), 0.134), ("fixture", h.cylinder_cloud, 1.0), ("planes200k", lambda: h.scene_planes(200_000, seed=2), None)):
    tgt = gen(); rng = np.random.default_rng(0)
    src = (tgt + rng.normal(0, 0.01, tgt.shape)).astype(np.float32)
    if hc is None: hc = 0.3
    for group, curve in ((64, "morton"), (64, "hilbert")):
        rows, tabn, pts, rs, ts = model(tgt, src, hc, group, curve)
        pc = lambda a: tuple(int(np.percentile(a, p)) for p in (50, 75, 90, 99))
        print(name, curve, "group", group, "rows p50/75/90/99", pc(rows), "tab", pc(tabn), "pts", pc(pts))
        for caps in ((128, 1024, 384), (256, 2048, 512), (512, 4096, 768)):
            ok = (rs <= caps[0]) & (ts <= caps[1]) & (pts <= caps[2])
            print("    caps", caps, "tile frac %.3f" % ok.mean())
