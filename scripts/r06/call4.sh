#!/bin/bash
O=gpurun_out/r06_c6; mkdir -p $O
timeout 600 python scripts/r06/rs_first.py > $O/parity.txt 2>&1; grep -c "'ok': False" $O/parity.txt; tail -1 $O/parity.txt
for st in 0 2 5 8; do echo "== RS=3 STAGGER=$st"; BPX_BWD_STAGGER=$st BPX_BWD_RS=3 timeout 300 python tests/bench_kernels.py bwd 2>&1 | grep "^bwd" | head -2 | cut -c90-; done > $O/bench.txt 2>&1
cat $O/bench.txt
