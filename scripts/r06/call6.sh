#!/bin/bash
O=gpurun_out/r06_c11; mkdir -p $O
for L in "" _wprio _wpriodma _dma; do echo "== lib$L"; BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd$L.so BPX_BWD_RS=3 timeout 300 python tests/bench_kernels.py bwd 2>&1 | grep "^bwd" | head -2 | cut -c90-; done > $O/bench.txt 2>&1
cat $O/bench.txt
