#!/bin/bash
O=gpurun_out/r05_call6
mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "zmarch" ) 2>&1 | grep -E "err=|passed|failed" | cut -c1-260 | head -20
for cap in 256 512 768; do
  m=$(( (cap << 8) | 2 ))
  echo "== zm, $cap workgroups"
  BPX_CONV_ZM=$m timeout 300 python tests/bench_kernels.py conv_fwd --dtype f16 --only 2,0 --reps 20 2>&1 | grep conv_fwd
  BPX_STAMP_SPLIT=1 BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd_zmstamps.so BPX_CONV_ZM=$m python scripts/zm_stamps.py 2 2>&1 | grep -v amdgpu.ids
done > $O/wg_sweep.txt 2>&1
cat $O/wg_sweep.txt
