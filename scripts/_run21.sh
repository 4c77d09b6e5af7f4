timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "deferred_norm or network_cfg2 or reproducible or mixed_training_follows or resunet_matches or groupnorm or gn" 2>&1 | tail -4
for i in 1 2; do
for V in "BPX_NBF_DEFER=0" "BPX_NBF_DEFER=1"; do
  echo "== $V"
  env $V timeout 400 python bench.py --mode train --no-cpu-baseline --no-bf16-record --steps 30 2>&1 | grep "train record" | cut -c60-140
done; done
