#!/bin/bash
# why do short runs read slower?  same call with different warm-up / step counts
O=gpurun_out/r05_call24
mkdir -p $O
run() { timeout 300 python bench.py --mode train --steps $1 --warmup $2 --no-cpu-baseline --no-bf16-record --no-launch-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps $1 warmup $2 ms_per_step %.4f' % d['ms_per_step'])"; }
for rep in 1 2; do
  run 10 5; run 10 60; run 20 5; run 20 60; run 80 5
done > $O/steps_warmup.txt 2>&1
cat $O/steps_warmup.txt
