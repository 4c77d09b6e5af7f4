#!/bin/bash
# A/B: 4x8x16 dgrad tile at four workgroups per CU (libbiapy_amd.so) against three (libbiapy_amd_alt.so)
O=gpurun_out/r2f; mkdir -p $O
python tests/bench_kernels.py conv_dgrad > $O/dgrad_occ4.txt 2>&1
python bench.py --mode train --no-cpu-baseline > $O/train_occ4.json 2> $O/err4.txt
python -m pytest tests -m gpu -q -p no:cacheprovider -k "conv or network" > $O/pytest_occ4.txt 2>&1
cp biapy_amd/libbiapy_amd.so /tmp/main.so; cp biapy_amd/libbiapy_amd_alt.so biapy_amd/libbiapy_amd.so
python tests/bench_kernels.py conv_dgrad > $O/dgrad_occ3.txt 2>&1
python bench.py --mode train --no-cpu-baseline > $O/train_occ3.json 2> $O/err3.txt
cp /tmp/main.so biapy_amd/libbiapy_amd.so
paste $O/dgrad_occ4.txt $O/dgrad_occ3.txt | cut -c1-200
cat $O/train_occ4.json | cut -c1-300; cat $O/train_occ3.json | cut -c1-300; tail -3 $O/pytest_occ4.txt
