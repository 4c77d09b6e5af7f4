#!/bin/bash
O=gpurun_out/r05_call17
mkdir -p $O
for rep in 1 2; do
  for tag in o1p3 tree o2p2 o2p4; do
    unset BPX_LIB_PATH
    [ $tag != tree ] && export BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd_$tag.so
    echo "== $tag rep $rep"
    python bench.py --breakdown --graph off --mode infer 2>/dev/null | grep "sum =\|convT3d_k2s2_fwd (2, 4, 64"
  done
done > $O/breakdown_ab.txt 2>&1
cat $O/breakdown_ab.txt
