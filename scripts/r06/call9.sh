#!/bin/bash
# lean-kernel branch-free epilogue: parity subset + kernel A/B (tree vs HEAD) + train/infer step A/B
O=gpurun_out/r06_c14; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv or zmarch or chunk_planar or pool or shuffle or saturat" > $O/tests.txt 2>&1; tail -2 $O/tests.txt
for L in "" _ab; do echo "== lib$L"; BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd$L.so timeout 300 python tests/bench_kernels.py conv_fwd 2>&1 | grep -E "^(conv|fwd|dgrad)" | head -40; done > $O/kern.txt 2>&1
for L in "" _ab "" _ab; do echo "== lib$L"; BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd$L.so timeout 300 python bench.py --mode train --steps 40 --warmup 5 --no-cpu-baseline --no-launch-events --no-bf16-record 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d.get('infer_ms_per_step'))"; done > $O/ab.txt 2>&1
cat $O/ab.txt
