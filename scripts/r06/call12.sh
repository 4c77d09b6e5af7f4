#!/bin/bash
O=gpurun_out/r06_c17; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "zmarch or pool or conv3d_kernel" 2>&1 | tail -2
for L in "" _ab "" _ab; do echo "== lib$L"; BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd$L.so python tests/bench_kernels.py conv_fwd --dtype f16 --only 0,1,2 --reps 20 2>&1 | grep conv_fwd; done
for L in "" _ab "" _ab; do BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd$L.so python bench.py --mode infer --steps 40 --warmup 8 --no-cpu-baseline --no-launch-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib$L infer ms_per_step %.4f' % d['ms_per_step'])"; done
