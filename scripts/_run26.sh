timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "norm_pool_head or network_cfg2_arch or resunet_matches or sliding_window_cfg3" 2>&1 | tail -3
for i in 1 2; do
for V in "BPX_C1_PERSIST=0" "BPX_C1_PERSIST=1280" "BPX_C1_PERSIST=2048" "BPX_C1_PERSIST=768"; do
  echo "== $V"
  env $V timeout 400 python bench.py --mode train --no-cpu-baseline --no-bf16-record --steps 30 2>&1 | grep "train record" | cut -c60-140
  [ $i = 1 ] && env $V timeout 300 python bench.py --mode train --breakdown --no-cpu-baseline 2>&1 | grep -E "conv3d_c1_fwd"
done; done
