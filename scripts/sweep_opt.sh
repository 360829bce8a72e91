#!/bin/bash
# GPU box: per-iteration times of the product library for several values of one context option.  usage: sweep_opt.sh key v1 v2 ...
cd "$(dirname "$0")/.."
key=$1; shift
for wl in c4_corridor_1m c2_cylinder_100k c3_pk01_200k; do
for rep in 1 2; do
for v in "$@"; do echo "== $wl $key=$v"; python scripts/iter_times.py $wl $key=$v 2>&1 | grep -v amdgpu | sed -n 2p; done; done; done
