#!/bin/bash
# z-march epilogue: buffer stores with out-of-range offsets (tree) against predicated stores behind an explicit wait (BPX_ZM_BUFST=0 build)
O=gpurun_out/r05_call21
mkdir -p $O
AB=$PWD/biapy_amd/libbiapy_amd_nobufst.so
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "zmarch or fused_maxpool" ) > $O/tests.txt 2>&1
tail -4 $O/tests.txt
for rep in 1 2 3; do
  echo "== buffer stores"; timeout 200 python tests/bench_kernels.py conv_fwd --dtype f16 --only 0,1,2 --reps 30 2>&1 | grep conv_fwd
  echo "== predicated stores"; BPX_LIB_PATH=$AB timeout 200 python tests/bench_kernels.py conv_fwd --dtype f16 --only 0,1,2 --reps 30 2>&1 | grep conv_fwd
done > $O/micro_ab.txt 2>&1
cat $O/micro_ab.txt
run() { env $1 timeout 300 python bench.py --mode $2 --steps 40 --warmup 8 --no-cpu-baseline --no-bf16-record --no-launch-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2 ms_per_step %.4f' % d['ms_per_step'])"; }
for rep in 1 2 3; do
  run BPX_LIB_PATH=$AB infer; run BPX_X=0 infer
done > $O/step_ab.txt 2>&1
for rep in 1 2; do
  run BPX_LIB_PATH=$AB train; run BPX_X=0 train
done >> $O/step_ab.txt 2>&1
cat $O/step_ab.txt
python bench.py --mode sliding --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:v for k,v in d.items() if 'check' in k or 'ms' in k})" > $O/sliding.txt 2>&1; cat $O/sliding.txt
