cd /root/repo
V=$PWD/dcreg_amd/lib/libdcreg_hip_unbatched.so
for i in 1 2; do
for lib in "" "$V"; do echo "== c4 ${lib:-product}"; DCREG_LIB=$lib python scripts/iter_times.py c4_corridor_1m 2>&1 | grep -v amdgpu | head -2 | cut -c1-200; done
done
for lib in "" "$V"; do echo "== c1 ${lib:-product}"; DCREG_LIB=$lib python scripts/iter_times.py c1_fixture_7562 2>&1 | grep -v amdgpu | sed -n 2p; done
for lib in "" "$V"; do echo "== c3 ${lib:-product}"; DCREG_LIB=$lib python scripts/iter_times.py c3_pk01_200k 2>&1 | grep -v amdgpu | sed -n 2p; done
