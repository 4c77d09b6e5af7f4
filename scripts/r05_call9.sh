#!/bin/bash
O=gpurun_out/r05_call9
mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "zmarch" ) 2>&1 | grep -E "err=|passed|failed" | cut -c1-260 | head -20
for rep in 1 2; do
  echo "== zm (all instances)"; BPX_CONV_ZM=2 BPX_CONV_ZM_MASK=7 timeout 300 python tests/bench_kernels.py conv_fwd --dtype f16 --only 1 --reps 20 2>&1 | grep conv_fwd
  echo "== lean"; BPX_CONV_ZM=0 timeout 300 python tests/bench_kernels.py conv_fwd --dtype f16 --only 1 --reps 20 2>&1 | grep conv_fwd
done > $O/micro.txt 2>&1
cat $O/micro.txt
run() { BPX_CONV_ZM=$1 BPX_CONV_ZM_MASK=$2 timeout 300 python bench.py --mode $3 --steps 40 --warmup 8 --no-cpu-baseline --no-bf16-record --no-launch-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('zm $1 mask $2 $3 ms_per_step %.4f' % d['ms_per_step'])"; }
for rep in 1 2 3; do run 1 5 train; run 1 7 train; done > $O/step_ab.txt 2>&1
for rep in 1 2; do run 1 5 infer; run 1 7 infer; done >> $O/step_ab.txt 2>&1
cat $O/step_ab.txt
