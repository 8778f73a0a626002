#!/bin/bash
# Register / LDS / occupancy table of every kernel of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
# Usage: scripts/kernel_resources.sh transformer-recommenders_amd/csrc/gemm.hip [extra hipcc flags]
src=$1; shift
cd "$(dirname "$src")" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I/root/repo/include "$@" -c "$(basename "$src")" -o /dev/null \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re, sys
rows, cur = [], {}
for line in sys.stdin:
    m = re.search(r"remark: [^ ]+ +(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|SGPRs): (.*?) \[-Rpass", line)
    if not m: 
        m = re.search(r"(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
        if not m: continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": v}; rows.append(cur)
    else:
        cur[k.split(" ")[0]] = v
print("VGPR AGPR scratch occ LDS  name")
for r in rows:
    print(r.get("VGPRs","?"), r.get("AGPRs","?"), r.get("ScratchSize","?"), r.get("Occupancy","?"), r.get("LDS","?"), r["name"][:150])
'
