#!/bin/bash
# round 5: first-layer kernels with the hi + lo image split at staging, transposed-conv forward with the XCD swizzle - parity tests, then same-box
# A/B against the previous commit's library (biapy_amd/libbiapy_amd_ab.so) / BPX_CONVT_SWZ=0
O=gpurun_out/r05_call12
mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "pointwise_and_transposed or first_layer or convT or zmarch" ) > $O/tests.txt 2>&1
tail -5 $O/tests.txt
AB=$PWD/biapy_amd/libbiapy_amd_ab.so
for rep in 1 2 3; do
  echo "== new"; timeout 200 python tests/bench_kernels.py c1 --dtype f16 --reps 30 2>&1 | grep "c1_"
  echo "== old"; BPX_LIB_PATH=$AB timeout 200 python tests/bench_kernels.py c1 --dtype f16 --reps 30 2>&1 | grep "c1_"
done > $O/c1_ab.txt 2>&1
cat $O/c1_ab.txt
run() { env $1 timeout 300 python bench.py --mode $2 --steps 40 --warmup 8 --no-cpu-baseline --no-bf16-record --no-launch-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2 ms_per_step %.4f' % d['ms_per_step'])"; }
for rep in 1 2; do
  run BPX_LIB_PATH=$AB infer; run BPX_CONVT_SWZ=0 infer; run BPX_CONVT_SWZ=1 infer
done > $O/step_ab.txt 2>&1
for rep in 1 2; do
  run BPX_LIB_PATH=$AB train; run BPX_CONVT_SWZ=0 train; run BPX_CONVT_SWZ=1 train
done >> $O/step_ab.txt 2>&1
cat $O/step_ab.txt
python bench.py --breakdown --graph off --mode infer 2>/dev/null | grep "convT3d_k2s2_fwd\|c1_fwd" > $O/breakdown_new.txt
BPX_CONVT_SWZ=0 python bench.py --breakdown --graph off --mode infer 2>/dev/null | grep "convT3d_k2s2_fwd\|c1_fwd" > $O/breakdown_noswz.txt
echo new; cat $O/breakdown_new.txt; echo noswz; cat $O/breakdown_noswz.txt
python bench.py --mode sliding --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:v for k,v in d.items() if 'check' in k or 'ms' in k})" > $O/sliding.txt 2>&1; cat $O/sliding.txt
