O=gpurun_out/r04d; mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider ) > $O/gpu_tests.txt 2>&1
timeout 400 python bench.py --mode train --no-cpu-baseline --steps 30 > $O/train.json 2> $O/train.err
tail -5 $O/gpu_tests.txt; grep "train record" $O/train.err
