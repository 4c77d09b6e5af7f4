timeout 300 python tests/bench_kernels.py rcan 2>&1 | grep "graph replay"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "block_activations or fused_conv_backward or norm_act or rcan or resunetpp_against or unet" 2>&1 | tail -4
timeout 400 python bench.py --mode train --no-cpu-baseline --no-bf16-record --steps 30 2>&1 | grep "train record"
