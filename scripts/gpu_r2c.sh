#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2c
mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q --maxfail=15 --tb=short -p no:cacheprovider ) > $O/pytest.txt 2>&1
grep -n "passed\|failed" $O/pytest.txt | tail -3
python tests/bench_kernels.py merge > $O/merge_crop.txt 2>&1
cat $O/merge_crop.txt
