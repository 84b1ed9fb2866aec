#!/bin/bash
# Print VGPR/SGPR/LDS/scratch/occupancy per kernel (hipcc -Rpass-analysis=kernel-resource-usage).
cd "$(dirname "$0")"
for f in k_blend_fwd k_backward k_project k_binning; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -c $f.hip -o /tmp/$f.res.o \
     -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
cur=None
for line in sys.stdin:
    m=re.search(r'Function Name: (\S+)',line)
    if m: cur={'name':m.group(1)[:36]}; continue
    m=re.search(r'remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+) \[-Rpass',line)
    if m and cur is not None:
        cur[m.group(1).strip()]=m.group(2)
        if m.group(1).startswith('LDS Size'):
            print('%-38s vgpr=%s sgpr=%s scratch=%s occ=%s lds=%s' % (cur['name'],cur.get('VGPRs'),cur.get('TotalSGPRs'),cur.get('ScratchSize'),cur.get('Occupancy'),cur.get('LDS Size'))); cur=None
"
done
