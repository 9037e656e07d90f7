#!/bin/bash
# Print VGPR/AGPR/LDS/occupancy per kernel of one csrc/*.hip file (compiler view, no GPU needed).
# usage: tools/kernel_resources.sh pswin_attn.hip
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
CSRC="$ROOT/panoswintransformerobjectdetection_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -ffp-contract=off -I"$ROOT/include" -I"$CSRC" \
  -c "$CSRC/$1" -o /tmp/_kr.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
cur=None
for line in sys.stdin:
    m=re.search(r'Function Name: (\S+)',line)
    if m: cur=m.group(1); print(); print(cur[:80],end=' | ')
    for key in ['VGPRs:','AGPRs:','TotalSGPRs:','Occupancy [waves/SIMD]:','LDS Size [bytes/block]:','VGPRs Spill:','ScratchSize [bytes/lane]:']:
        if key in line and cur:
            print(key.split()[0], line.split(key)[1].split('[')[0].strip(), end=' | ')
print()
"
