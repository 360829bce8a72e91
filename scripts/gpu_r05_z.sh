#!/bin/bash
# round 5, call Z: the one pair of call Y's engine hunt that drifts from the oracle (seed 62, case 1: 29 effective points) - with every pass off,
# and with the parity plane fit
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
echo "== passes off"; DCREG_FUZZ_OPTS="advance=0 team_pass=0" timeout 300 python scripts/fuzz_engine.py 2 62 2>&1 | tail -4
echo "== parity fit"; DCREG_FUZZ_OPTS="fast_plane_fit=0" timeout 300 python scripts/fuzz_engine.py 2 62 2>&1 | tail -4
echo "== parity fit, passes off, certificates off"; DCREG_FUZZ_OPTS="fast_plane_fit=0 advance=0 team_pass=0 use_certificates=0" timeout 300 python scripts/fuzz_engine.py 2 62 2>&1 | tail -4
