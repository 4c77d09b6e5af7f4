#!/bin/bash
O=gpurun_out/r05_call4
mkdir -p $O
run() { BPX_CONV_ZM=$1 BPX_CONV_ZM_MASK=$2 timeout 300 python bench.py --mode train --steps 40 --warmup 8 --no-cpu-baseline --no-bf16-record --no-launch-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('zm $1 mask $2 train ms_per_step %.4f' % d['ms_per_step'])"; }
for rep in 1 2; do
  run 0 7; run 1 4; run 1 1; run 1 2; run 1 5; run 1 7
done > $O/mask_ab.txt 2>&1
cat $O/mask_ab.txt
# clocks while the step loops: sampled in the background
for zm in 1 0; do
  ( for i in $(seq 1 40); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.25; done ) > $O/smi_zm$zm.txt &
  SP=$!
  BPX_CONV_ZM=$zm timeout 300 python bench.py --mode train --steps 600 --warmup 8 --no-cpu-baseline --no-bf16-record --no-launch-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('zm $zm 600 steps ms_per_step %.4f' % d['ms_per_step'])"
  wait $SP
done > $O/long.txt 2>&1
cat $O/long.txt; tail -5 $O/smi_zm1.txt; tail -5 $O/smi_zm0.txt
