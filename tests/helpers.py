"""Shared test helpers: golden CSV parsing and comparison utilities; the clouds, scenes and poses come from dcreg_amd.scenes."""
import csv
import gzip
import os

import numpy as np

from dcreg_amd.scenes import *                    # noqa: F401,F403  (read_pcd_xyz, cylinder_cloud, scene_*, pose6d_matrix, deg2rad, ...)
from dcreg_amd.scenes import FIXTURE_PCD          # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
REPO = os.path.dirname(HERE)


def read_csv_rows(path):
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rt") as f:
        return list(csv.DictReader(f))


def golden_rows(family, fname, method=None):
    rows = read_csv_rows(os.path.join(GOLDEN, family, fname))
    if method is not None:
        rows = [r for r in rows if r["Method"] == method]
    return rows


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))
