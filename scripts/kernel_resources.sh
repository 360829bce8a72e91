#!/bin/bash
# Registers / scratch / LDS / occupancy of every kernel of libdcreg_hip.so, as the compiler reports them (no GPU needed).
# usage: scripts/kernel_resources.sh [extra hipcc flags]
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -c dcreg_amd/csrc/device/context.hip -o /tmp/dcreg_res.o \
    -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c '
import re, sys, subprocess
rows, cur = [], None
for line in sys.stdin:
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m: continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}; rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
print("%-52s %5s %5s %8s %7s %5s" % ("kernel", "VGPR", "SGPR", "scratch", "LDS", "occ"))
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n).replace("dcreg::", "").replace("void ", "")
    print("%-52s %5s %5s %8s %7s %5s" % (n[:52], r.get("VGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))
'
