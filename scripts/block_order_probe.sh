cd /root/repo
for o in "block_order=0" "block_order=1" "block_order=2" "block_order=2 xcd_chunk=0" "block_order=0 xcd_chunk=1" "block_order=2 xcd_chunk=1"; do echo "== $o"; python scripts/iter_times.py c4_corridor_1m $o 2>&1 | grep -v amdgpu | head -2 | cut -c1-200; done
