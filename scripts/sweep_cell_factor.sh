#!/bin/bash
cd "$(dirname "$0")/.."
for wl in c4_corridor_1m c2_cylinder_100k c3_pk01_200k; do
for cf in 1.4 1.7 2.0 2.4 2.8; do echo "== $wl cell_factor=$cf"; python scripts/iter_times.py $wl cell_factor=$cf 2>&1 | grep -v amdgpu | sed -n 2p; done; done
