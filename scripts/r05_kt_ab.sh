#!/bin/bash
# kernel-trace A/B of the train step: z-march on / off (same box, same call)
O=$PWD/gpurun_out/r05_kt
mkdir -p $O
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
for zm in 1 0; do
  BPX_CONV_ZM=$zm timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt$zm -o train -- python $ROOT/bench.py --mode train --steps 20 --warmup 3 --no-cpu-baseline --no-launch-events --no-bf16-record > $O/kt$zm.log 2>&1
  cp $(find $O/kt$zm -name "train_kernel_stats.csv" | head -1) $O/train_kernel_stats_zm$zm.csv
  rm -rf $O/kt$zm
done
head -12 $O/train_kernel_stats_zm1.csv | cut -c1-200
echo; head -12 $O/train_kernel_stats_zm0.csv | cut -c1-200
