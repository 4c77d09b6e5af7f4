#!/bin/bash
O=gpurun_out/r06_c12; mkdir -p $O
timeout 600 python scripts/r06/rs_first.py > $O/parity.txt 2>&1; grep -c "'ok': False" $O/parity.txt; tail -1 $O/parity.txt
for L in "" _nodb; do echo "== lib$L"; BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd$L.so timeout 300 python tests/bench_kernels.py bwd 2>&1 | grep "^bwd" | head -2 | cut -c90-; done > $O/bench.txt 2>&1
echo "== serial"; BPX_BWD_RS=0 timeout 300 python tests/bench_kernels.py bwd 2>&1 | grep "^bwd" | head -2 | cut -c90- >> $O/bench.txt
cat $O/bench.txt
BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd_stamps.so python scripts/bwd_rs_stamps.py 128 48 2>&1 | grep -v amdgpu.ids | tee $O/stamps.txt
