#!/bin/bash
O=gpurun_out/r05_call5
mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "zmarch" ) 2>&1 | tail -3
for rep in 1 2; do
  echo "== zm: 16-byte stores + 2-load image";  BPX_CONV_ZM=2 timeout 300 python tests/bench_kernels.py conv_fwd --dtype f16 --only 0,1,2 --reps 20 2>&1 | grep conv_fwd
  echo "== zm: 8-byte stores";                  BPX_CONV_ZM=2 BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd_zmst8.so timeout 300 python tests/bench_kernels.py conv_fwd --dtype f16 --only 0,1,2 --reps 20 2>&1 | grep conv_fwd
  echo "== lean";                                BPX_CONV_ZM=0 timeout 300 python tests/bench_kernels.py conv_fwd --dtype f16 --only 0,1,2 --reps 20 2>&1 | grep conv_fwd
done > $O/micro_ab.txt 2>&1
cat $O/micro_ab.txt
for k in 2 0; do BPX_STAMP_SPLIT=1 BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd_zmstamps.so BPX_CONV_ZM=2 python scripts/zm_stamps.py $k 2>&1 | grep -v amdgpu.ids; done > $O/stamps.txt
cat $O/stamps.txt
