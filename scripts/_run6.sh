O=gpurun_out/r04f; mkdir -p $O
for v in 0 4096 32768; do BPX_SIDE_VPS=$v timeout 400 python bench.py --mode train --no-cpu-baseline --no-bf16-record --steps 30 > $O/train_side$v.json 2> $O/train_side$v.err; grep "train record" $O/train_side$v.err; done
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "bit_reproducible or cfg2_arch or network_against_reference or graphed_train_step or saturate or train_one_epoch" 2>&1 | tail -5
