#!/bin/bash
# GPU box: FETCH_SIZE / WRITE_SIZE of kernels that move known bytes in k_lin's access patterns -> gpurun_out/fetch_calib/ (summarised by
# scripts/microbench/fetch_calib_summary.py into profiles/rNN_fetch_calibration.{md,json})
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/fetch_calib; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
/opt/rocm/bin/hipcc -O2 -Wno-unused-value --offload-arch=gfx950 $R/scripts/microbench/fetch_calib.hip -o /tmp/fetch_calib || exit 1
/tmp/fetch_calib > $O/known.json
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- /tmp/fetch_calib > $O/fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- /tmp/fetch_calib > $O/write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $O/rdreq -- /tmp/fetch_calib > $O/rdreq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $O/wrreq -- /tmp/fetch_calib > $O/wrreq.log 2>&1
python $R/scripts/microbench/fetch_calib_summary.py ${1:-r04}
