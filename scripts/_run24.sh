timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "streaming_decoder_input" 2>&1 | tail -6
for i in 1 2; do
for V in "BPX_PW_STREAM=0" "BPX_PW_STREAM=1"; do
  echo "== $V"
  env $V timeout 400 python bench.py --mode train --no-cpu-baseline --no-bf16-record --steps 30 2>&1 | grep "train record" | cut -c60-140
done; done
timeout 300 python bench.py --mode train --breakdown --no-cpu-baseline 2>&1 | grep -E "conv1x1_fwd_split|wgrad_db2 .*, 1\)|wgrad_db2 .*, 1, " | head -8
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "network_cfg2 or reproducible or mixed_training_follows or resunet_matches" 2>&1 | tail -3
