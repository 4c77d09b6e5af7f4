timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "streaming_transposed or convT or streaming_shortcut" 2>&1 | tail -5
for i in 1 2; do
for V in "BPX_WGRAD_K1=3" "BPX_WGRAD_K1=1"; do
  echo "== $V"
  env $V timeout 400 python bench.py --mode train --no-cpu-baseline --no-bf16-record --steps 30 2>&1 | grep "train record" | cut -c60-140
  [ $i = 1 ] && env $V timeout 300 python bench.py --mode train --breakdown --no-cpu-baseline 2>&1 | grep -E "convT3d_k2s2_wgrad "
done; done
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "network_cfg2_arch or reproducible or mixed_training_follows" 2>&1 | tail -3
