#!/bin/bash
O=gpurun_out/r05_call10
mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "resunetpp" ) > $O/tests_pp.txt 2>&1
tail -4 $O/tests_pp.txt
for rep in 1 2; do
  for f in 1 0; do
    BPX_BWD_FUSED=$f timeout 300 python bench.py --arch resunetpp --batch 4 --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg4 fused-bwd $f ms_per_step %.3f' % d['ms_per_step'])"
  done
done > $O/cfg4_ab.txt 2>&1
cat $O/cfg4_ab.txt
