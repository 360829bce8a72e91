#!/bin/bash
# GPU box: build and run the VALU issue microbenchmark; JSON on stdout (committed as profiles/rNN_valu_microbench.json)
set -e
cd "$(dirname "$0")/../.."
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -Wno-unused-value scripts/microbench/valu_issue.hip -o /tmp/valu_issue
/tmp/valu_issue
