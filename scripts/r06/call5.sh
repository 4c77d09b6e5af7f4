#!/bin/bash
O=gpurun_out/r06_c9; mkdir -p $O
for L in "" _dma _spread; do echo "== lib$L"; BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd$L.so BPX_BWD_RS=3 timeout 300 python tests/bench_kernels.py bwd 2>&1 | grep "^bwd" | head -2 | cut -c90-; done > $O/bench.txt 2>&1
cat $O/bench.txt
BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd_spread.so python scripts/r06/rs_first.py 2>&1 | tail -1
