#!/bin/bash
# GPU box: per-iteration times of the product library vs dcreg_amd/lib/libdcreg_hip_base.so (a copy of an earlier build), interleaved.
cd "$(dirname "$0")/.."
V=${AB_LIB:-$PWD/dcreg_amd/lib/libdcreg_hip_base.so}
for wl in c4_corridor_1m c2_cylinder_100k c3_pk01_200k c1_fixture_7562; do
for i in 1 2; do
for lib in "" "$V"; do echo "== $wl ${lib:+base}"; DCREG_LIB=$lib python scripts/iter_times.py $wl "$@" 2>&1 | grep -v amdgpu | sed -n 1,2p | cut -c1-200; done
done; done
