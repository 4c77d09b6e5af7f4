#!/bin/bash
O=gpurun_out/r05_call20
mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "first_layer or one_k_step or norm_pool_head" ) > $O/tests.txt 2>&1
tail -6 $O/tests.txt
for rep in 1 2 3; do
  echo "== buffer"; timeout 200 python tests/bench_kernels.py c1 --dtype f16 --reps 30 2>&1 | grep "c1_fwd"
  echo "== pointer"; BPX_C1_PERSIST=$((2048 + (1<<30))) timeout 200 python tests/bench_kernels.py c1 --dtype f16 --reps 30 2>&1 | grep "c1_fwd"
done > $O/c1_ab.txt 2>&1
cat $O/c1_ab.txt
run() { env $1 timeout 300 python bench.py --mode $2 --steps 40 --warmup 8 --no-cpu-baseline --no-bf16-record --no-launch-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2 ms_per_step %.4f' % d['ms_per_step'])"; }
for rep in 1 2 3; do
  run BPX_C1_PERSIST=$((2048 + (1<<30))) infer; run BPX_C1_PERSIST=2048 infer
done > $O/step_ab.txt 2>&1
cat $O/step_ab.txt
python bench.py --mode sliding --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:v for k,v in d.items() if 'check' in k or 'ms' in k})" > $O/sliding.txt 2>&1; cat $O/sliding.txt
