#!/bin/bash
O=gpurun_out/r2h; mkdir -p $O
( time python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=10 --tb=short ) > $O/pytest.txt 2>&1; tail -25 $O/pytest.txt | cut -c1-300
python bench.py --mode train --no-cpu-baseline > $O/train.json 2> $O/train.err; cut -c1-330 $O/train.json
python bench.py --arch resunetpp --batch 4 --steps 5 --warmup 2 > $O/pp_graph.json 2> $O/pp_graph.err; cut -c1-330 $O/pp_graph.json
python bench.py --arch resunetpp --batch 4 --breakdown --graph off > $O/pp_breakdown.txt 2>&1; head -12 $O/pp_breakdown.txt | cut -c1-160
