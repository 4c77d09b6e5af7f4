#!/bin/bash
O=gpurun_out/r05_call3
mkdir -p $O
for zm in 1 0; do
  BPX_CONV_ZM=$zm timeout 300 python bench.py --breakdown --graph off --mode train > $O/breakdown_zm$zm.txt 2>/dev/null
done
for rep in 1 2 3; do
  for zm in 0 1; do
    BPX_CONV_ZM=$zm timeout 300 python bench.py --mode train --steps 60 --warmup 10 --no-cpu-baseline --no-bf16-record --no-launch-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('zm $zm train ms_per_step', d['ms_per_step'])"
  done
done > $O/step_ab.txt 2>&1
cat $O/step_ab.txt
rocm-smi --showpower --showclocks 2>/dev/null | head -30 > $O/smi.txt
