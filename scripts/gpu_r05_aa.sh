#!/bin/bash
# round 5, call AA: the small-frame pass's lower bound ("team_pass_min_frac": the pass runs when the last completed launch searched at least this
# share of the frame), registration, alternating on one box
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for rep in 1 2; do
  for f in 0.5 0.3 0.15 0.05; do
    echo "== team_pass_min_frac=$f (rep $rep)"; timeout 200 python scripts/reg_probe.py team_pass_min_frac=$f 2>&1 | sed -n '1,2p;5p'
  done
done
