#!/bin/bash
# VGPRs / AGPRs / scratch / occupancy of every kernel of csrc/*.hip, one line per kernel, sorted: the table is committed as
# profiles/kernel_resources.txt and diffed after a change (`bash scripts/kernel_resources.sh | diff profiles/kernel_resources.txt -`): a refactor
# of a shared kernel template can silently cost another instance a wave per SIMD (round 3: a persistent loop added for the transposed-conv
# forward took the 48-column 1x1x1 GEMM of the same template from 130 to 170 VGPRs and from 402 to 598 us).
cd "$(dirname "$0")/../biapy_amd/csrc"
for F in tiling conv3d_igemm conv3d_lean conv3d_zmarch wgrad bwd_fused pointwise elementwise prepost; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fPIC -c $F.hip -o /tmp/kres_$F.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
cur=None; rows=[]
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur=[m.group(1),0,0,0,0,0]; rows.append(cur)
    for i,k in ((1,'VGPRs'),(2,'AGPRs'),(3,r'ScratchSize \[bytes/lane\]'),(4,r'Occupancy \[waves/SIMD\]'),(5,r'LDS Size \[bytes/block\]')):
        m=re.search(r'remark:\s+'+k+r': (\d+)',l)
        if m and cur: cur[i]=int(m.group(1))
for r in sorted(rows): print('$F vgpr %3d agpr %3d scratch %4d occ %d lds %6d  %s'%(r[1],r[2],r[3],r[4],r[5],r[0]))
" &
done
wait
