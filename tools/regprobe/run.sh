#!/bin/bash
# usage: tools/regprobe/run.sh [-DNAME=VALUE ...]   -> per-kernel VGPRs / scratch / occupancy / LDS of the hot f32 kernels (offline, no GPU)
cd "$(dirname "$0")/../.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -ffp-contract=fast -fno-slp-vectorize -ffast-math -fno-finite-math-only \
  -Wno-unused-value --cuda-device-only -c ${PROBE:-tools/regprobe/probe.hip} -o /tmp/regprobe.o -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c '
import re, subprocess, sys
cur = None
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().replace("smac::", "").split("(")[0]}
        continue
    if "error" in line and "remark" not in line:
        print(line.rstrip()); continue
    for key, tag in (("VGPRs", "VGPR"), ("AGPRs", "AGPR"), (r"ScratchSize \[bytes/lane\]", "scratch"), (r"Occupancy \[waves/SIMD\]", "occ"), (r"LDS Size \[bytes/block\]", "LDS")):
        m = re.search(rf"remark:\s+{key}: (\d+)", line)
        if m and cur is not None:
            cur[tag] = m.group(1)
            if tag == "LDS" and "smac" not in cur["name"][:0]:
                print("{name:44s} VGPR {VGPR:>3s} AGPR {AGPR:>3s} scratch {scratch:>4s} occ {occ} LDS {LDS}".format(**cur))
'
