#!/bin/bash
# Round-2 GPU call B: suite again (row kernels with batched loads, fixed tests), merge/crop A-B, occupancy sweeps of the lean conv
# and the windowed wgrad kernels on the 128^3 / 64^3 layers.
export TMPDIR=/tmp
O=gpurun_out/r2b
mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 --tb=short -p no:cacheprovider ) > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
python tests/bench_kernels.py merge > $O/merge_crop.txt 2>&1
cat $O/merge_crop.txt
for occ in 0 3 4 5 6; do
  for i in 0 1 2 3 4; do BPX_OCC=$occ python tests/bench_kernels.py conv_fwd --only $i 2>&1 | grep conv_fwd | sed "s/^/occ=$occ /"; done
  for i in 0 1 3 4; do BPX_OCC=$occ python tests/bench_kernels.py conv_dgrad --only $i 2>&1 | grep conv_dgrad | sed "s/^/occ=$occ /"; done
done > $O/conv_occ.txt 2>&1
cat $O/conv_occ.txt
# wgrad: workgroups in percent of the co-resident capacity the launch bounds assume (bits 8.. of the hook; bit 0 = tr16 operands)
for pct in 100 125 133 150; do
  for i in 0 1 3 4; do BPX_WGRAD=$((1 + pct * 256)) python tests/bench_kernels.py wgrad --only $i 2>&1 | grep "^wgrad" | sed "s/^/pct=$pct /"; done
done > $O/wgrad_pct.txt 2>&1
cat $O/wgrad_pct.txt
