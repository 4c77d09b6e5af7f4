#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2e
mkdir -p $O
( BPX_CONV_DBG=32 timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "conv3d or network_against or cfg2" --tb=short -p no:cacheprovider ) > $O/pytest_pf.txt 2>&1
grep -n "passed\|failed" $O/pytest_pf.txt | tail -3
( timeout 900 python -m pytest tests -m gpu -q --maxfail=15 --tb=short -p no:cacheprovider ) > $O/pytest.txt 2>&1
grep -n "passed\|failed" $O/pytest.txt | tail -3
for dbg in 0 16 32 48; do
  for i in 0 1 2 3 4; do BPX_CONV_DBG=$dbg python tests/bench_kernels.py conv_fwd --only $i 2>&1 | grep conv_fwd | sed "s/^/dbg=$dbg /"; done
  for i in 0 1 3 4; do BPX_CONV_DBG=$dbg python tests/bench_kernels.py conv_dgrad --only $i 2>&1 | grep conv_dgrad | sed "s/^/dbg=$dbg /"; done
done > $O/conv_dbg.txt 2>&1
cat $O/conv_dbg.txt
python tests/bench_kernels.py merge 2>&1 | grep "row kernels"
