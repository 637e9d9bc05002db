#!/bin/bash
# register / spill / scratch summary of the kernels of dvo_fused.hip (device-only compile, ~40 s): tools/fused_resources.sh [extra -D flags]
cd $(dirname $0)/../rgbd_odometry_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math --cuda-device-only --no-gpu-bundle-output -c dvo_fused.hip -o /tmp/dvo_fused_dev.o \
  -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c '
import re, sys
cur = None
for l in sys.stdin:
    m = re.search(r"Function Name: (\S+)", l)
    if m: cur = m.group(1); vals = {}
    for k in ("VGPRs", "AGPRs", "ScratchSize \[bytes/lane\]", "VGPR Spill", "SGPRs", "LDS Size \[bytes/block\]", "Occupancy \[waves/SIMD\]"):
        m = re.search(r"remark: .*    %s: (\d+)" % k, l)
        if m: vals[k.split()[0]] = int(m.group(1))
    if cur and "LDS" in l:
        short = re.sub(r"^_ZN3dvo19", "", cur)[:34]
        print("%-36s %s" % (short, " ".join("%s=%d" % kv for kv in vals.items())))
        cur = None
'
