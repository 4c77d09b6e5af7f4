#!/bin/bash
O=gpurun_out/r05_call8
mkdir -p $O
run() { BPX_CONV_ZM=$1 BPX_CONV_ZM_MASK=$2 timeout 300 python bench.py --mode $3 --steps 40 --warmup 8 --no-cpu-baseline --no-bf16-record --no-launch-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('zm $1 mask $2 $3 ms_per_step %.4f' % d['ms_per_step'])"; }
for rep in 1 2 3; do
  run 0 7 train; run 1 7 train; run 1 5 train
done > $O/step_ab.txt 2>&1
for rep in 1 2; do run 0 7 infer; run 1 7 infer; done >> $O/step_ab.txt 2>&1
cat $O/step_ab.txt
