#!/bin/bash
# Lists every kernel of the given csrc file that the compiler gave scratch memory (a spilling kernel ran up to 2x slower inside the network than alone:
# every 16-bit instance is meant to compile without).  bash scripts/check_scratch.sh conv3d_lean.hip [extra hipcc flags]
cd "$(dirname "$0")/../biapy_amd/csrc"
F=$1; shift
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fPIC "$@" -c $F -o /tmp/check_scratch.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
cur=None; rows=[]
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur=[m.group(1),0,0,0]; rows.append(cur)
    for i,k in ((1,'VGPRs'),(2,r'ScratchSize \[bytes/lane\]'),(3,r'Occupancy \[waves/SIMD\]')):
        m=re.search(r'remark:\s+'+k+r': (\d+)',l)
        if m and cur: cur[i]=int(m.group(1))
bad=[r for r in rows if r[2]>0]
print(len(rows),'kernels,',len(bad),'with scratch')
for r in bad: print('  scratch %4d B  vgpr %3d occ %d  %s'%(r[2],r[1],r[3],r[0][:150]))
"
