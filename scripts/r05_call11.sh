#!/bin/bash
# role-split z-march kernel (conv3_zs_kernel): bit equality with the lean kernel, then micro and step A/B against conv3_zm_kernel
O=gpurun_out/r05_call11
mkdir -p $O
( BPX_CONV_ZS=1 timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "zmarch" ) > $O/tests_zs.txt 2>&1
tail -15 $O/tests_zs.txt
for rep in 1 2 3; do
  for zs in 0 1; do
    echo "== zs $zs rep $rep"
    BPX_CONV_ZS=$zs timeout 200 python tests/bench_kernels.py conv_fwd --dtype f16 --only 2 --reps 30 2>&1 | grep -v "^$"
  done
done > $O/micro_ab.txt 2>&1
cat $O/micro_ab.txt
run() { BPX_CONV_ZS=$1 timeout 300 python bench.py --mode $2 --steps 40 --warmup 8 --no-cpu-baseline --no-bf16-record --no-launch-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('zs $1 $2 ms_per_step %.4f' % d['ms_per_step'])"; }
for rep in 1 2; do run 0 train; run 1 train; done > $O/step_ab.txt 2>&1
for rep in 1 2; do run 0 infer; run 1 infer; done >> $O/step_ab.txt 2>&1
cat $O/step_ab.txt
