#!/bin/bash
O=gpurun_out/r05_call19
mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "pointwise or convT or transposed or resunet_cfg2" ) > $O/tests.txt 2>&1
tail -4 $O/tests.txt
for rep in 1 2; do
  for tag in k0 tree p3; do
    unset BPX_LIB_PATH BPX_CONVT_K1
    [ $tag = k0 ] && export BPX_CONVT_K1=0
    [ $tag = p3 ] && export BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd_p3.so
    [ $tag = p1 ] && export BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd_p1.so
    echo "== $tag rep $rep"
    python bench.py --breakdown --graph off --mode infer 2>/dev/null | grep "sum =\|convT3d_k2s2_fwd"
  done
done > $O/breakdown_ab.txt 2>&1
unset BPX_LIB_PATH BPX_CONVT_K1
cat $O/breakdown_ab.txt
run() { env $1 timeout 300 python bench.py --mode $2 --steps 40 --warmup 8 --no-cpu-baseline --no-bf16-record --no-launch-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2 ms_per_step %.4f' % d['ms_per_step'])"; }
for rep in 1 2 3; do
  run BPX_CONVT_K1=0 infer; run BPX_CONVT_K1=1 infer
done > $O/step_ab.txt 2>&1
for rep in 1 2; do
  run BPX_CONVT_K1=0 train; run BPX_CONVT_K1=1 train
done >> $O/step_ab.txt 2>&1
cat $O/step_ab.txt
python bench.py --mode sliding --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:v for k,v in d.items() if 'check' in k or 'ms' in k})" > $O/sliding.txt 2>&1; cat $O/sliding.txt
