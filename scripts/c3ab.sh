cd /root/repo
V=$PWD/dcreg_amd/lib/libdcreg_hip_unbatched.so
for i in 1 2 3; do
for lib in "" "$V"; do echo "== c3 ${lib:-product}"; DCREG_LIB=$lib python scripts/iter_times.py c3_pk01_200k 2>&1 | grep -v amdgpu | sed -n 1,2p | cut -c1-260; done
done
