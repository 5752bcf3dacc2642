#!/bin/bash
# Dev: registers / spills / scratch of every onf_x32_kernel instance in a built library (default: the product library).
L=${1:-$(cd "$(dirname "$0")/../.." && pwd)/pytorch-motion-planner_amd/nfopp/lib/libnfopp_hip.so}
T=$(mktemp -d); cp "$L" $T/lib.so; cd $T
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so >/dev/null 2>&1
for f in *gfx950*; do /opt/rocm/lib/llvm/bin/llvm-readelf --notes "$f" | python3 -c "
import sys,re
t=sys.stdin.read()
for m in re.finditer(r'\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)', t, re.S):
    if '${2:-onf_x32_kernel}' in m.group(1): print(m.group(1)[:72], 'scratch',m.group(2),'sgpr',m.group(3),'vgpr',m.group(4),'spill',m.group(5))
"; done
rm -rf $T
