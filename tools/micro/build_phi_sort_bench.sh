#!/bin/bash
# Build tools/micro/bin/phi_sort_bench and print one line of register usage per phi_sort_kernel instantiation.
cd "$(dirname "$0")/../.." || exit 1
mkdir -p tools/micro/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -Rpass-analysis=kernel-resource-usage "$@" \
  tools/micro/phi_sort_bench.hip -o tools/micro/bin/phi_sort_bench 2> /tmp/phi_sort_build.log || { grep -E "error" -A3 /tmp/phi_sort_build.log | head -40; exit 1; }
python3 - <<'PY'
import re
name = None; rows = {}
for line in open('/tmp/phi_sort_build.log'):
    m = re.search(r'Function Name: (\S+)', line)
    if m: name = m.group(1); rows[name] = {}
    for key in ('VGPRs', 'SGPRs Spill', 'VGPRs Spill', 'ScratchSize \[bytes/lane\]', 'Occupancy \[waves/SIMD\]'):
        m = re.search(r'remark:\s+' + key + r': (\d+)', line)
        if m and name: rows[name][key] = int(m.group(1))
for n, r in rows.items():
    if "phi_sort" in n:
        m = re.search(r"phi_sort_kernelILi(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)", n)
        tag = ("sort K=%s TP=%s ABL=%s PF=%s TS=%s" % m.groups()) if m else n[:40]
        print('%-28s VGPR %3d  spillV %3d spillS %3d scratch %4d occ %d' % (tag, r.get('VGPRs', -1), r.get('VGPRs Spill', -1), r.get('SGPRs Spill', -1), r.get('ScratchSize \\[bytes/lane\\]', -1), r.get('Occupancy \\[waves/SIMD\\]', -1)))
PY
