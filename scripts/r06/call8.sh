#!/bin/bash
# full GPU test suite on the tree + train-step A/B of the role-split mask
O=gpurun_out/r06_c13; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
for m in 0 1 3 0 3; do echo "== BPX_BWD_RS=$m"; BPX_BWD_RS=$m timeout 300 python bench.py --mode train --steps 40 --warmup 5 --no-cpu-baseline --no-launch-events --no-bf16-record 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])"; done > $O/ab.txt 2>&1
cat $O/ab.txt
