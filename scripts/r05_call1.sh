#!/bin/bash
# Round 5, first GPU call: the z-marching forward kernel (bit-equality with the lean kernel, micro-benchmarks A/B, step A/B) and the new parity tests.
O=gpurun_out/r05_call1
mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "zmarch or explicit_act or dropout_backward or chunked_prediction or fused_adam or kernel_variants or fused_maxpool" ) > $O/tests_new.txt 2>&1
tail -5 $O/tests_new.txt
for rep in 1 2; do
  echo "== zm on (RH4, image early)";  timeout 300 python tests/bench_kernels.py conv_fwd --dtype f16 --only 0,1,2 --reps 20 2>&1 | grep conv_fwd
  echo "== zm off (lean)";             BPX_CONV_ZM=0 timeout 300 python tests/bench_kernels.py conv_fwd --dtype f16 --only 0,1,2 --reps 20 2>&1 | grep conv_fwd
  echo "== zm on (RH8, image late)";   BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd_zmrh8.so timeout 300 python tests/bench_kernels.py conv_fwd --dtype f16 --only 0,1,2 --reps 20 2>&1 | grep conv_fwd
done > $O/micro_ab.txt 2>&1
cat $O/micro_ab.txt
for rep in 1 2; do
  BPX_CONV_ZM=1 timeout 300 python bench.py --mode train --steps 30 --warmup 5 --no-cpu-baseline --no-bf16-record --no-launch-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('zm on  train ms_per_step', d['ms_per_step'])"
  BPX_CONV_ZM=0 timeout 300 python bench.py --mode train --steps 30 --warmup 5 --no-cpu-baseline --no-bf16-record --no-launch-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('zm off train ms_per_step', d['ms_per_step'])"
done > $O/step_ab.txt 2>&1
cat $O/step_ab.txt
BPX_CONV_ZM=1 timeout 300 python bench.py --mode infer --steps 30 --warmup 5 --no-cpu-baseline --no-launch-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('zm on  infer', d.get('ms_per_step'), d.get('infer',{}).get('ms_per_step'))" >> $O/step_ab.txt 2>&1
BPX_CONV_ZM=0 timeout 300 python bench.py --mode infer --steps 30 --warmup 5 --no-cpu-baseline --no-launch-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('zm off infer', d.get('ms_per_step'), d.get('infer',{}).get('ms_per_step'))" >> $O/step_ab.txt 2>&1
tail -2 $O/step_ab.txt
( time timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "mixed_training_follows" ) > $O/tests_dice.txt 2>&1
tail -5 $O/tests_dice.txt; cat gpurun_out/diag_values.txt 2>/dev/null
