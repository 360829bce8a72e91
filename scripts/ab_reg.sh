#!/bin/bash
# A/B of the small-frame registration (scripts/reg_probe.py) on ONE box: this tree against a worktree under _ab/<name>, alternating.
# usage: scripts/ab_reg.sh <name> [rounds] [reg_probe options of THIS tree ...]
cd "$(dirname "$0")/.."
NAME=$1; ROUNDS=${2:-3}; shift 2
for r in $(seq 1 $ROUNDS); do
  (cd _ab/$NAME && python scripts/reg_probe.py 2>/dev/null | head -3 | sed "s/^/$NAME: /")
  python scripts/reg_probe.py "$@" 2>/dev/null | head -3 | sed "s/^/new: /"
done
