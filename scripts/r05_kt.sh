#!/bin/bash
O=$PWD/gpurun_out/r05_kt2
mkdir -p $O
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o train -- python $ROOT/bench.py --mode train --steps 20 --warmup 3 --no-cpu-baseline --no-launch-events --no-bf16-record > $O/kt.log 2>&1
cp $(find $O/kt -name "train_kernel_stats.csv" | head -1) $O/train_kernel_stats.csv
rm -rf $O/kt
head -14 $O/train_kernel_stats.csv | cut -c1-160
