#!/bin/bash
# VGPRs / scratch / occupancy / LDS of every kernel in tracker_kernels.hip (developer aid)
cd "$(dirname "$0")/../direct_stereo_slam_amd/csrc" || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -S --cuda-device-only \
  -Rpass-analysis=kernel-resource-usage ${1:-tracker_kernels.hip} -o /dev/null 2>&1 |
  python3 -c '
import re, sys
cur = None
for l in sys.stdin:
    m = re.search(r"remark: +(.*?) \[-Rpass", l)
    if not m: continue
    s = m.group(1).strip()
    if s.startswith("Function Name:"):
        cur = s.split(":", 1)[1].strip(); vals = {}
    elif cur and ":" in s:
        k, v = s.split(":", 1); vals[k.strip()] = v.strip()
        if k.strip().startswith("LDS Size"):
            print("%-110s VGPR %4s scratch %3s occ %s LDS %s" % (cur[:110], vals.get("VGPRs"), vals.get("ScratchSize [bytes/lane]"), vals.get("Occupancy [waves/SIMD]"), v.strip()))
'
