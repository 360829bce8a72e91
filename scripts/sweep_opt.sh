#!/bin/bash
# GPU box: mean iteration time of whole runs (scripts/run_probe.py) for several values of one context option.
# usage: sweep_opt.sh key v1 v2 ...   [WORKLOADS="c4_corridor_1m c2_cylinder_100k" to restrict]
cd "$(dirname "$0")/.."
key=$1; shift
for wl in ${WORKLOADS:-c4_corridor_1m c2_cylinder_100k c3_pk01_200k}; do
for v in "$@"; do echo -n "$wl $key=$v: "; python scripts/run_probe.py $wl $key=$v 2>&1 | grep -v amdgpu | tail -1; done; done
