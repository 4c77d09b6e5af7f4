#!/bin/bash
O=gpurun_out/r2i; mkdir -p $O
python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "planar" --tb=short > $O/pytest_planar.txt 2>&1; tail -15 $O/pytest_planar.txt | cut -c1-250
( time python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=10 --tb=short ) > $O/pytest.txt 2>&1; tail -12 $O/pytest.txt | cut -c1-250
python bench.py --mode train --no-cpu-baseline > $O/train.json 2> $O/train.err; cut -c1-330 $O/train.json
BPX_PLANAR_CAT=0 python bench.py --mode train --no-cpu-baseline > $O/train_interleaved.json 2> $O/train2.err; cut -c1-330 $O/train_interleaved.json
python bench.py --mode infer --no-cpu-baseline > $O/infer.json 2> $O/infer.err; cut -c1-330 $O/infer.json
BPX_PLANAR_CAT=0 python bench.py --mode infer --no-cpu-baseline > $O/infer_interleaved.json 2> $O/infer2.err; cut -c1-330 $O/infer_interleaved.json
python bench.py --breakdown --graph off --mode train > $O/breakdown_train.txt 2> /dev/null
