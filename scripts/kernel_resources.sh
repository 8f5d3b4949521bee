#!/bin/bash
# Registers, LDS, scratch and occupancy of every kernel of one source file, as the compiler reports them.
# Usage: scripts/kernel_resources.sh lm_kernels
f=${1:-lm_kernels}
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -munsafe-fp-atomics -Iinclude -c stereo_visual_odometry_amd/csrc/$f.hip -o /tmp/$f.res.o -Rpass-analysis=kernel-resource-usage 2>&1 |
python3 -c '
import re, sys, subprocess
rows = {}; cur = None
for line in sys.stdin:
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip(); rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1); rows[cur][k.strip()] = v.strip()
for name, r in rows.items():
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
    print("%-44s VGPR %4s AGPR %3s SGPR %3s scratch %5s LDS %6s occ %s" % (dem[-44:], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))
'
