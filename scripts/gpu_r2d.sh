#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2d
mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q --maxfail=15 --tb=short -p no:cacheprovider ) > $O/pytest.txt 2>&1
grep -n "passed\|failed" $O/pytest.txt | tail -3
python tests/bench_kernels.py merge > $O/merge_crop.txt 2>&1
cat $O/merge_crop.txt
for b in 2 4; do timeout 600 python bench.py --arch resunetpp --steps 5 --warmup 2 --batch $b > $O/bench_resunetpp_b$b.json 2> $O/bench_resunetpp_b$b.err; tail -c 900 $O/bench_resunetpp_b$b.json; tail -2 $O/bench_resunetpp_b$b.err; done
