timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "first_layer_weight or norm_pool_head or network_cfg2_arch or reproducible" 2>&1 | tail -3
for i in 1 2; do
for L in ab tree; do
  if [ "$L" = ab ]; then export BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd_ab.so; else unset BPX_LIB_PATH; fi
  echo "== lib: $L"
  timeout 400 python bench.py --mode train --no-cpu-baseline --no-bf16-record --steps 30 2>&1 | grep "train record" | cut -c60-140
  [ $i = 1 ] && timeout 300 python bench.py --mode train --breakdown --no-cpu-baseline 2>&1 | grep -E "c1_wgrad"
done; done
