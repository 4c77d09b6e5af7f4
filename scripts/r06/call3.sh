#!/bin/bash
O=gpurun_out/r06_c10; mkdir -p $O
timeout 600 python scripts/r06/rs_first.py > $O/parity.txt 2>&1; grep -v "'ok': True" $O/parity.txt | tail -8
for m in 0 3; do echo "== BPX_BWD_RS=$m"; BPX_BWD_RS=$m timeout 300 python tests/bench_kernels.py bwd 2>&1 | grep "^bwd" | head -2; done > $O/bench.txt 2>&1
cat $O/bench.txt
BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd_stamps.so python scripts/bwd_rs_stamps.py 128 48 > $O/stamps.txt 2>&1; BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd_stamps.so python scripts/bwd_rs_stamps.py 128 16 >> $O/stamps.txt 2>&1
grep -v amdgpu.ids $O/stamps.txt
