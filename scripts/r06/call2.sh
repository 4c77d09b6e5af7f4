#!/bin/bash
O=$PWD/gpurun_out/r06_c2; mkdir -p $O
export TMPDIR=/tmp; ROOT=$PWD; cd /tmp
for m in 0 3; do
BPX_BWD_RS=$m timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt$m -o b -- python $ROOT/tests/bench_kernels.py bwd > $O/kt$m.log 2>&1
cp $(find $O/kt$m -name "b_kernel_stats.csv" | head -1) $O/stats$m.csv; rm -rf $O/kt$m
grep "conv3_bwd" $O/stats$m.csv | cut -c1-150
done
