#!/usr/bin/env python
"""Writes the seeded PK01 stand-in pair that configs/icp_pk01.yaml points at (BASELINE config 3):

    gpurun_out/pk01_standin/target_prior_map.pcd            ~200 k-point planar prior map (ground + poles + kerbs)
    gpurun_out/pk01_standin/parkinglot_raw_2415_frame.pcd   one 8 k-point frame in the sensor frame (sigma = 2 cm)

The reference's real pair is not in its repository (README.md:69, Google Drive).  With the real files in that folder the
same YAML runs unchanged.  Generator: dcreg_amd/scenes.py scene_parkinglot (numpy default_rng, fixed seed).
Usage: python scripts/make_pk01_standin.py [out_dir] [n_frame]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dcreg_amd import scenes as h  # noqa: E402


def main(out_dir=None, n_frame=8000):
    out_dir = out_dir or os.path.join(ROOT, "gpurun_out", "pk01_standin")
    tgt, src = h.scene_parkinglot(n_frame=n_frame, frame_range=30.0 if n_frame <= 20000 else 100.0)
    h.write_pcd_xyzi(os.path.join(out_dir, "target_prior_map.pcd"), tgt)
    h.write_pcd_xyzi(os.path.join(out_dir, "parkinglot_raw_2415_frame.pcd"), src)
    print("wrote %d-point map and %d-point frame to %s" % (len(tgt), len(src), out_dir))
    return tgt, src


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None, int(sys.argv[2]) if len(sys.argv) > 2 else 8000)
