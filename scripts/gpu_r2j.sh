#!/bin/bash
O=gpurun_out/r2j; mkdir -p $O
for pad in 0 256 69888 1114368; do
  BPX_PLANE_PAD=$pad python bench.py --mode train --no-cpu-baseline > $O/train_$pad.json 2> $O/err_$pad.txt
  BPX_PLANE_PAD=$pad python bench.py --breakdown --graph off --mode train > $O/bd_$pad.txt 2> /dev/null
  echo "pad $pad: $(python -c "import json;print(json.load(open('$O/train_$pad.json'))['ms_per_step'])")"
  grep "convT3d_k2s2_fwd (1, 4, 64\|convT3d_k2s2_fwd (1, 4, 32\|conv1x1_fwd_split (1, 4, 'C16'\|fwd_pool (1, 4, 128\|maxpool3d_bwd (1, 4, 128\|conv3d_fwd (1, 4, 128, 128, 128, 'C48'" $O/bd_$pad.txt | cut -c1-120
done
python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "planar or golden or sliding" --tb=short 2>&1 | tail -3
