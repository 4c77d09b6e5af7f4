#!/bin/bash
O=gpurun_out/r2g; mkdir -p $O
python -m pytest tests -m gpu -q -p no:cacheprovider -k "resunetpp" > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
python bench.py --arch resunetpp --batch 4 --steps 5 --warmup 2 > $O/pp_graph.json 2> $O/pp_graph.err; tail -3 $O/pp_graph.err; cut -c1-400 $O/pp_graph.json
python bench.py --arch resunetpp --batch 4 --steps 5 --warmup 2 --graph off > $O/pp_eager.json 2> $O/pp_eager.err; cut -c1-400 $O/pp_eager.json
python bench.py --arch resunetpp --batch 4 --steps 5 --warmup 2 --force-ddp > $O/pp_dp1.json 2> $O/pp_dp1.err; tail -3 $O/pp_dp1.err; cut -c1-400 $O/pp_dp1.json
