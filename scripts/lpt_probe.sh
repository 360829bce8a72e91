#!/bin/bash
cd "$(dirname "$0")/.."
for o in "lpt=0" "lpt=1" "lpt=2" "lpt=1 lpt_spread=1.2" "lpt=1 lpt_spread=3" "lpt=0" "lpt=1"; do echo "== $o"; python scripts/iter_times.py c4_corridor_1m $o 2>&1 | grep -v amdgpu | head -2 | cut -c1-210; done
echo "== c3 lpt=0"; python scripts/iter_times.py c3_pk01_200k lpt=0 2>&1 | grep -v amdgpu | sed -n 2p
echo "== c3 lpt=1"; python scripts/iter_times.py c3_pk01_200k lpt=1 2>&1 | grep -v amdgpu | sed -n 2p
python -m pytest tests/test_gpu_configs.py -m gpu -q -k "c4 or c2" 2>&1 | tail -2
