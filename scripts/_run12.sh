O=gpurun_out/r04j; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "fused_conv_backward or block_activations" 2>&1 | tail -8
timeout 300 python tests/bench_kernels.py bwd --reps 10 2>&1 | grep bwd
timeout 300 python tests/bench_kernels.py rcan 2>&1 | grep -v "Warning\|run_backward\|amdgpu" | head -4
for v in 3 1; do BPX_FUSED_BITS=$v timeout 400 python bench.py --mode train --no-cpu-baseline --no-bf16-record --steps 30 2>&1 | grep "train record"; done
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "overlapped or two_process or bit_reproducible or cfg2_arch or network_against_reference or train_one_epoch or graphed" 2>&1 | tail -5
