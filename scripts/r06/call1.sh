#!/bin/bash
# first GPU contact of the role-split fused backward: parity, then timing (serial fused vs role-split), then stamps
O=gpurun_out/r06_c1; mkdir -p $O
timeout 600 python scripts/r06/rs_first.py > $O/parity.txt 2>&1; tail -3 $O/parity.txt
for m in 0 1 2 3; do echo "== BPX_BWD_RS=$m"; BPX_BWD_RS=$m timeout 300 python tests/bench_kernels.py bwd 2>&1 | grep "^bwd" | head -2; done > $O/bench.txt 2>&1
cat $O/bench.txt
