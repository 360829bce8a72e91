"""Shared test helpers: fixture loading, golden CSV parsing, synthetic scenes."""
import csv
import gzip
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
REPO = os.path.dirname(HERE)


def read_pcd_xyz(path):
    """Minimal PCD v0.7 reader (binary / ascii, float32 fields) -> float32 [n,3]."""
    with open(path, "rb") as f:
        fields, sizes, counts, npts, data_kind = [], [], [], 0, None
        while True:
            line = f.readline().decode("ascii", "replace").strip()
            if line.startswith("FIELDS"):
                fields = line.split()[1:]
            elif line.startswith("SIZE"):
                sizes = [int(v) for v in line.split()[1:]]
            elif line.startswith("COUNT"):
                counts = [int(v) for v in line.split()[1:]]
            elif line.startswith("POINTS"):
                npts = int(line.split()[1])
            elif line.startswith("DATA"):
                data_kind = line.split()[1]
                break
        if not counts:
            counts = [1] * len(fields)
        rec = sum(s * c for s, c in zip(sizes, counts))
        if data_kind == "binary":
            raw = np.frombuffer(f.read(rec * npts), dtype=np.uint8).reshape(npts, rec)
            off = {}
            o = 0
            for name, s, c in zip(fields, sizes, counts):
                off[name] = o
                o += s * c
            cols = [raw[:, off[k]:off[k] + 4].copy().view(np.float32)[:, 0] for k in ("x", "y", "z")]
            return np.stack(cols, axis=1).astype(np.float32)
        txt = np.loadtxt(f, dtype=np.float64).reshape(npts, -1)
        ix = [fields.index(k) for k in ("x", "y", "z")]
        return txt[:, ix].astype(np.float32)


def cylinder_cloud():
    return read_pcd_xyz(os.path.join(GOLDEN, "cylinder_7562.pcd"))


def read_csv_rows(path):
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rt") as f:
        return list(csv.DictReader(f))


def golden_rows(family, fname, method=None):
    rows = read_csv_rows(os.path.join(GOLDEN, family, fname))
    if method is not None:
        rows = [r for r in rows if r["Method"] == method]
    return rows


def deg2rad(d):
    return d * np.pi / 180.0


# initial poses of the two committed trace families (complete_log.txt of each run)
RELEASE_INIT = dict(x=0.01, y=0.01, z=0.01, roll=0.0, pitch=0.0, yaw=0.0)
PAPER_INIT = dict(x=0.2, y=0.8, z=0.5, roll=deg2rad(0.1), pitch=deg2rad(0.1), yaw=deg2rad(2.0))


def pose6d_matrix(x, y, z, roll, pitch, yaw):
    """T * Rz(yaw) * Ry(pitch) * Rx(roll)  (utils.hpp:452-460)."""
    cr, sr, cp, sp, cy, sy = np.cos(roll), np.sin(roll), np.cos(pitch), np.sin(pitch), np.cos(yaw), np.sin(yaw)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = [x, y, z]
    return T


# ------------------------------------------------------------------ synthetic scenes (seeded)
def scene_cylinder(n, seed=0, radius=40.0, height=20.0, noise=0.0):
    """Cylinder wall + floor disk, like the reference's simulated scene, jittered (no exact ties)."""
    rng = np.random.default_rng(seed)
    n_wall = n // 2
    n_floor = n - n_wall
    th = rng.uniform(0, 2 * np.pi, n_wall)
    z = rng.uniform(0, height, n_wall)
    wall = np.stack([radius * np.cos(th), radius * np.sin(th), z], 1)
    r = radius * np.sqrt(rng.uniform(0, 1, n_floor))
    th2 = rng.uniform(0, 2 * np.pi, n_floor)
    floor = np.stack([r * np.cos(th2), r * np.sin(th2), rng.normal(0, 1e-3, n_floor)], 1)
    pts = np.concatenate([wall, floor], 0)
    if noise > 0:
        pts = pts + rng.normal(0, noise, pts.shape)
    rng.shuffle(pts)
    return pts.astype(np.float32)


def scene_corridor(n, seed=0, length=200.0, width=4.0, height=3.0, noise=0.005):
    """Two walls + floor + ceiling along x: translational degeneracy along the axis."""
    rng = np.random.default_rng(seed)
    per = 2 * (width + height)
    u = rng.uniform(0, per, n)
    x = rng.uniform(-length / 2, length / 2, n)
    y = np.empty(n)
    z = np.empty(n)
    a = u < width                                   # floor
    b = (u >= width) & (u < width + height)         # wall y=+w/2
    c = (u >= width + height) & (u < 2 * width + height)  # ceiling
    d = u >= 2 * width + height                     # wall y=-w/2
    y[a] = u[a] - width / 2; z[a] = 0.0
    y[b] = width / 2; z[b] = u[b] - width
    y[c] = width / 2 - (u[c] - width - height); z[c] = height
    y[d] = -width / 2; z[d] = height - (u[d] - 2 * width - height)
    pts = np.stack([x, y, z], 1) + rng.normal(0, noise, (n, 3))
    return pts.astype(np.float32)


def scene_planes(n, seed=0, extent=60.0, noise=0.01):
    """Ground plane + sparse vertical poles: X-Y-yaw weakly constrained (parking-lot stand-in)."""
    rng = np.random.default_rng(seed)
    n_poles = max(n // 20, 1)
    n_ground = n - n_poles
    g = np.stack([rng.uniform(-extent, extent, n_ground), rng.uniform(-extent, extent, n_ground),
                  np.zeros(n_ground)], 1)
    centers = rng.uniform(-extent, extent, (16, 2))
    k = rng.integers(0, 16, n_poles)
    ang = rng.uniform(0, 2 * np.pi, n_poles)
    p = np.stack([centers[k, 0] + 0.3 * np.cos(ang), centers[k, 1] + 0.3 * np.sin(ang),
                  rng.uniform(0, 4, n_poles)], 1)
    pts = np.concatenate([g, p], 0) + rng.normal(0, noise, (n, 3))
    rng.shuffle(pts)
    return pts.astype(np.float32)


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))
