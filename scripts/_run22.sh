timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "streaming_shortcut" 2>&1 | tail -6
timeout 300 python tests/bench_kernels.py k1 --reps 20 2>&1 | grep "wgrad k=1"
for i in 1 2; do
for V in "BPX_WGRAD_K1=0" "BPX_WGRAD_K1=1"; do
  echo "== $V"
  env $V timeout 400 python bench.py --mode train --no-cpu-baseline --no-bf16-record --steps 30 2>&1 | grep "train record" | cut -c60-140
done; done
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "network_cfg2 or reproducible or mixed_training_follows or resunet_matches" 2>&1 | tail -3
