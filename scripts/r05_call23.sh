#!/bin/bash
# transposed-conv weight gradient (tile kernel): next-tile register prefetch (default) against BPX_CT_PRE=0
O=gpurun_out/r05_call23
mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "pointwise_and_transposed or convT or transposed_conv" ) > $O/tests.txt 2>&1
tail -6 $O/tests.txt
for rep in 1 2 3; do
  for f in 1 0; do
    echo "== BPX_CT_PRE=$f rep $rep"
    BPX_CT_PRE=$f python bench.py --breakdown --graph off --mode train 2>/dev/null | grep "sum =\|convT3d_k2s2_wgrad (4"
  done
done > $O/breakdown_ab.txt 2>&1
cat $O/breakdown_ab.txt
run() { env $1 timeout 300 python bench.py --mode $2 --steps 40 --warmup 8 --no-cpu-baseline --no-bf16-record --no-launch-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2 ms_per_step %.4f' % d['ms_per_step'])"; }
for rep in 1 2 3; do
  run BPX_CT_PRE=0 train; run BPX_CT_PRE=1 train
done > $O/step_ab.txt 2>&1
cat $O/step_ab.txt
