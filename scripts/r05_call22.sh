#!/bin/bash
# first layer's weight gradient: buffer-addressed instance (counted waits: real two-tiles-ahead staging; coefficients through the scalar cache)
O=gpurun_out/r05_call22
mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "first_layer or norm_pool_head" ) > $O/tests.txt 2>&1
tail -6 $O/tests.txt
for rep in 1 2 3; do
  for f in 0 1; do
    echo "== pointer-instance flag $f rep $rep"
    BPX_C1_PERSIST=$((2048 + (f << 30))) python bench.py --breakdown --graph off --mode train 2>/dev/null | grep "sum =\|c1_wgrad_nb \|c1_fwd"
  done
done > $O/breakdown_ab.txt 2>&1
cat $O/breakdown_ab.txt
run() { env $1 timeout 300 python bench.py --mode $2 --steps 40 --warmup 8 --no-cpu-baseline --no-bf16-record --no-launch-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2 ms_per_step %.4f' % d['ms_per_step'])"; }
for rep in 1 2 3; do
  run BPX_C1_PERSIST=$((2048 + (1<<30))) train; run BPX_C1_PERSIST=2048 train
done > $O/step_ab.txt 2>&1
cat $O/step_ab.txt
