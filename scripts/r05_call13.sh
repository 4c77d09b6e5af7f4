#!/bin/bash
O=gpurun_out/r05_call13
mkdir -p $O
AB=$PWD/biapy_amd/libbiapy_amd_ab.so
for rep in 1 2; do
  for lib in new old; do
    echo "== $lib rep $rep"
    if [ $lib = old ]; then export BPX_LIB_PATH=$AB; else unset BPX_LIB_PATH; fi
    BPX_CONVT_SWZ=0 python bench.py --breakdown --graph off --mode train 2>/dev/null | grep "sum =\|c1_\|convT3d_k2s2_fwd (2, 4, 64"
  done
done > $O/breakdown_ab.txt 2>&1
unset BPX_LIB_PATH
cat $O/breakdown_ab.txt
run() { env $1 timeout 300 python bench.py --mode $2 --steps 40 --warmup 8 --no-cpu-baseline --no-bf16-record --no-launch-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2 ms_per_step %.4f' % d['ms_per_step'])"; }
for rep in 1 2 3; do
  run BPX_CONVT_SWZ=0 train; run BPX_LIB_PATH=$AB train
done > $O/step_ab.txt 2>&1
cat $O/step_ab.txt
