#!/bin/bash
O=gpurun_out/r06_c18; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wgrad or chunk_planar or reproduc or resunet_train or gradient" 2>&1 | tail -2
for L in "" _ab "" _ab; do echo "== lib$L"; BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd$L.so python tests/bench_kernels.py wgrad 2>&1 | grep "^wgrad" | head -12; done
for L in "" _ab "" _ab; do BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd$L.so python bench.py --mode train --feed device --steps 40 --warmup 8 --no-cpu-baseline --no-launch-events --no-bf16-record 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib$L train ms_per_step %.4f' % d['ms_per_step'])"; done
