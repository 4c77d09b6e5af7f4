timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "block_activations or class_head or norm_act or conv3d_backward or conv3d_forward or kernel_variants" 2>&1 | tail -12
timeout 300 python bench.py --mode train --no-cpu-baseline --no-bf16-record --steps 20 2>&1 | grep "train record"
