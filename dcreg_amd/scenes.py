"""Clouds and poses of the benchmark workloads: the reference's simulated-cylinder fixture (its own data file), seeded synthetic
scenes of the BASELINE configs (cylinder + floor, corridor, ground plane + poles, the PK01 parking-lot stand-in) and minimal PCD I/O.
Used by bench.py, the Monte-Carlo driver, scripts/ and the tests."""
import os

import numpy as np

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
# DCReg/dataset/icp_results/target_clouds.pcd of the reference (7562 points; its simulated experiment uses it as source AND target)
FIXTURE_PCD = os.path.join(DATA, "cylinder_7562.pcd")


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
    return read_pcd_xyz(FIXTURE_PCD)


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


# ------------------------------------------------------------------ PK01 stand-in (BASELINE config 3)
# The reference's parking-lot pair (config/icp_pk01.yaml:13-14: parkinglot_raw_2415_frame.pcd / target_prior_map.pcd) is
# not in the repository (Google-Drive link only, README.md:69).  Stand-in: a planar prior map around the yaml's ground-truth
# position (ground + sparse poles + a few low kerbs: X-Y-yaw weakly constrained, README.md:96) and one LiDAR frame cut out
# of it, expressed in the sensor frame by gt^-1 and perturbed by range noise.  Poses = the yaml's own numbers.
PK01_GT = dict(x=-109.831089, y=-395.052129, z=-1.025780, roll=deg2rad(-2.635654), pitch=deg2rad(-4.141885), yaw=deg2rad(117.972711))
PK01_INIT = dict(x=-109.979288618688, y=-395.174034820224, z=-0.900523132121, roll=deg2rad(-2.650863295637),
                 pitch=deg2rad(-2.836418366839), yaw=deg2rad(120.142832419935))


def scene_parkinglot(n_map=200_000, n_frame=8_000, seed=7, extent=45.0, frame_range=30.0, noise=0.02):
    """-> (target map [n_map,3] float32 in the map frame, source frame [n_frame,3] float32 in the sensor frame)."""
    rng = np.random.default_rng(seed)
    T_gt = pose6d_matrix(**PK01_GT)
    c = T_gt[:3, 3]
    n_pole, n_kerb = n_map // 25, n_map // 50
    n_ground = n_map - n_pole - n_kerb
    # ground: a gently tilted plane through the sensor's footprint
    gx, gy = rng.uniform(-extent, extent, n_ground), rng.uniform(-extent, extent, n_ground)
    ground = np.stack([c[0] + gx, c[1] + gy, c[2] - 1.8 + 0.01 * gx - 0.005 * gy + rng.normal(0, 0.01, n_ground)], 1)
    # lamp poles / tree trunks: 24 thin vertical cylinders
    pc = rng.uniform(-extent * 0.8, extent * 0.8, (24, 2))
    k = rng.integers(0, 24, n_pole)
    ang = rng.uniform(0, 2 * np.pi, n_pole)
    pz = rng.uniform(0, 4.0, n_pole)
    poles = np.stack([c[0] + pc[k, 0] + 0.15 * np.cos(ang), c[1] + pc[k, 1] + 0.15 * np.sin(ang),
                      c[2] - 1.8 + 0.01 * pc[k, 0] - 0.005 * pc[k, 1] + pz], 1)
    # kerbs: 3 low (0.4 m) vertical strips, 12 m long, random heading
    kc = rng.uniform(-extent * 0.7, extent * 0.7, (3, 2))
    kh = rng.uniform(0, np.pi, 3)
    j = rng.integers(0, 3, n_kerb)
    s = rng.uniform(-6, 6, n_kerb)
    kx, ky = kc[j, 0] + s * np.cos(kh[j]), kc[j, 1] + s * np.sin(kh[j])
    kerbs = np.stack([c[0] + kx, c[1] + ky, c[2] - 1.8 + 0.01 * kx - 0.005 * ky + rng.uniform(0, 0.4, n_kerb)], 1)
    tgt = np.concatenate([ground, poles, kerbs], 0) + rng.normal(0, 0.005, (n_map, 3))
    rng.shuffle(tgt)
    tgt = tgt.astype(np.float32)
    # one frame: map points within range of the sensor, in the sensor frame, with range noise
    d = np.linalg.norm(tgt[:, :2].astype(np.float64) - c[:2], axis=1)
    near = np.flatnonzero(d < frame_range)
    sel = rng.choice(near, size=min(n_frame, len(near)), replace=False)
    Rg, tg = T_gt[:3, :3], T_gt[:3, 3]
    body = (tgt[sel].astype(np.float64) - tg) @ Rg          # R^T (p - t)
    body += rng.normal(0, noise, body.shape)
    return tgt, body.astype(np.float32)


def scene_prior_map(n_map=50_000_000, n_frame=8_000, seed=11, extent=350.0, frame_range=30.0, noise=0.02):
    """A LARGE prior map around the PK01 ground-truth position and one LiDAR frame cut out of it - the regime the reference publishes its
    timings in (1 - 10 k-point frames against 53 - 241 M-point prior maps: README tables 6 / 7, results/long_duration experiments/table3_4).
    Map: 2 * extent metres square - tilted, gently undulating ground (~ 93 % of the points), building facades (vertical planes 8 m high,
    ~ 5 %), poles (~ 2 %); at the defaults ~ 100 points per square metre of ground.  Built in float32 blocks (a 50 M-point map is 600 MB;
    no global shuffle - the index sorts the points anyway).  -> (map [n_map,3] float32 in the map frame, frame [n_frame,3] float32 in the
    sensor frame); poses = PK01_GT / PK01_INIT."""
    rng = np.random.default_rng(seed)
    T_gt = pose6d_matrix(**PK01_GT)
    c = T_gt[:3, 3].astype(np.float32)
    n_wall, n_pole = n_map // 20, n_map // 50
    n_ground = n_map - n_wall - n_pole
    tgt = np.empty((n_map, 3), np.float32)
    e = np.float32(extent)

    def height(x, y):           # ground height above c.z - 1.8 at offsets (x, y) from the sensor's footprint
        return np.float32(0.01) * x - np.float32(0.005) * y + np.float32(0.15) * np.sin(x * np.float32(0.05)) * np.cos(y * np.float32(0.04))

    blk = 1 << 22
    for i0 in range(0, n_ground, blk):
        m = min(blk, n_ground - i0)
        gx = (rng.random(m, dtype=np.float32) * 2 - 1) * e
        gy = (rng.random(m, dtype=np.float32) * 2 - 1) * e
        tgt[i0:i0 + m, 0] = c[0] + gx
        tgt[i0:i0 + m, 1] = c[1] + gy
        tgt[i0:i0 + m, 2] = c[2] - np.float32(1.8) + height(gx, gy) + rng.standard_normal(m, dtype=np.float32) * np.float32(0.01)
    # facades: 64 vertical planes per 700 m x 700 m (the count grows with the map's area, so that a larger map is more of the same map and not
    # the same facades with more points on each), 20 - 60 m long, 8 m high, random position / heading (a dozen of them within the frame's range)
    area = max(1.0, (float(extent) / 350.0) ** 2)
    nf = int(round(64 * area))
    fc = ((rng.random((nf, 2), dtype=np.float32) * 2 - 1) * e * np.float32(0.9))
    fc[:12] = (rng.random((12, 2), dtype=np.float32) * 2 - 1) * np.float32(frame_range * 0.9)
    fh = rng.random(nf, dtype=np.float32) * np.float32(np.pi)
    fl = np.float32(20.0) + rng.random(nf, dtype=np.float32) * np.float32(40.0)
    j = rng.integers(0, nf, n_wall)
    sw = (rng.random(n_wall, dtype=np.float32) - np.float32(0.5)) * fl[j]
    wx, wy = fc[j, 0] + sw * np.cos(fh[j]), fc[j, 1] + sw * np.sin(fh[j])
    o = n_ground
    tgt[o:o + n_wall, 0] = c[0] + wx
    tgt[o:o + n_wall, 1] = c[1] + wy
    tgt[o:o + n_wall, 2] = c[2] - np.float32(1.8) + height(wx, wy) + rng.random(n_wall, dtype=np.float32) * np.float32(8.0)
    # poles: 400 thin vertical cylinders per 700 m x 700 m (two dozen within the frame's range)
    npc = int(round(400 * area))
    pc = (rng.random((npc, 2), dtype=np.float32) * 2 - 1) * e * np.float32(0.95)
    pc[:24] = (rng.random((24, 2), dtype=np.float32) * 2 - 1) * np.float32(frame_range * 0.9)
    k = rng.integers(0, npc, n_pole)
    ang = rng.random(n_pole, dtype=np.float32) * np.float32(2 * np.pi)
    o += n_wall
    tgt[o:o + n_pole, 0] = c[0] + pc[k, 0] + np.float32(0.15) * np.cos(ang)
    tgt[o:o + n_pole, 1] = c[1] + pc[k, 1] + np.float32(0.15) * np.sin(ang)
    tgt[o:o + n_pole, 2] = c[2] - np.float32(1.8) + height(pc[k, 0], pc[k, 1]) + rng.random(n_pole, dtype=np.float32) * np.float32(4.0)
    tgt[n_ground:] += rng.standard_normal((n_wall + n_pole, 3), dtype=np.float32) * np.float32(0.005)
    # one frame: map points within range of the sensor, in the sensor frame, with range noise
    dx, dy = tgt[:, 0] - c[0], tgt[:, 1] - c[1]
    near = np.flatnonzero(dx * dx + dy * dy < np.float32(frame_range * frame_range))
    sel = rng.choice(near, size=min(n_frame, len(near)), replace=False)
    Rg, tg = T_gt[:3, :3], T_gt[:3, 3]
    body = (tgt[sel].astype(np.float64) - tg) @ Rg          # R^T (p - t)
    body += rng.normal(0, noise, body.shape)
    return tgt, body.astype(np.float32)


def write_pcd_xyzi(path, xyz):
    """Binary PCD v0.7, fields x y z intensity (float32), like pcl::io::savePCDFileBinary<PointXYZI>."""
    xyz = np.asarray(xyz, np.float32).reshape(-1, 3)
    n = len(xyz)
    rec = np.zeros((n, 4), np.float32)
    rec[:, :3] = xyz
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as f:
        f.write(("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\n"
                 "COUNT 1 1 1 1\nWIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA binary\n" % (n, n)).encode("ascii"))
        f.write(rec.tobytes())
