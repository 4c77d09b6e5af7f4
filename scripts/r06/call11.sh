#!/bin/bash
O=gpurun_out/r06_c16; mkdir -p $O
timeout 600 python scripts/r06/rs_first.py > $O/parity.txt 2>&1; grep -c "'ok': False" $O/parity.txt; tail -1 $O/parity.txt
for L in "" _ab "" _ab; do echo "== lib$L"; BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd$L.so timeout 300 python tests/bench_kernels.py bwd 2>&1 | grep "^bwd" | head -2 | cut -c1-22,88-; done > $O/bench.txt 2>&1
cat $O/bench.txt
for c in 48 16; do BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd_stamps.so python scripts/bwd_rs_stamps.py 128 $c 2>&1 | grep -v amdgpu.ids; done | tee $O/stamps.txt
