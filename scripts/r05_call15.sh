#!/bin/bash
# transposed-conv forward: bias set up in front of the block loop (+ operand ring depth 3 = tree, depth 1 = pd1) against the previous commit (ab)
O=gpurun_out/r05_call15
mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "pointwise or convT or conv1x1 or transposed or resunet_cfg2 or unet" ) > $O/tests.txt 2>&1
tail -4 $O/tests.txt
for rep in 1 2; do
  for tag in ab pd1 tree; do
    if [ $tag = tree ]; then unset BPX_LIB_PATH; else export BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd_$tag.so; fi
    echo "== $tag rep $rep"
    python bench.py --breakdown --graph off --mode train 2>/dev/null | grep "sum =\|convT3d_k2s2_\|conv1x1_fwd"
  done
done > $O/breakdown_ab.txt 2>&1
unset BPX_LIB_PATH
cat $O/breakdown_ab.txt
run() { env $1 timeout 300 python bench.py --mode $2 --steps 40 --warmup 8 --no-cpu-baseline --no-bf16-record --no-launch-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2 ms_per_step %.4f' % d['ms_per_step'])"; }
for rep in 1 2 3; do
  run BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd_ab.so infer; run BPX_X=0 infer
done > $O/step_ab.txt 2>&1
for rep in 1 2 3; do
  run BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd_ab.so train; run BPX_X=0 train
done >> $O/step_ab.txt 2>&1
cat $O/step_ab.txt
python bench.py --mode sliding --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:v for k,v in d.items() if 'check' in k or 'ms' in k})" > $O/sliding.txt 2>&1; cat $O/sliding.txt
