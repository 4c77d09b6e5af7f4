timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "adam_step_kernel or mixed_training_follows or train_loop or graphed or data_parallel" 2>&1 | tail -5
for i in 1 2; do
for V in "BPX_FUSED_ADAM=0" "BPX_FUSED_ADAM=1"; do
  echo "== $V"
  env $V timeout 400 python bench.py --mode train --no-cpu-baseline --no-bf16-record --steps 30 2>&1 | grep "train record" | cut -c60-140
done; done
