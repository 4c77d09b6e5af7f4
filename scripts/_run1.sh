mkdir -p gpurun_out/r04a
O=gpurun_out/r04a
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "fused_conv_backward" 2>&1 | tail -25 > $O/tests_fused.txt
timeout 300 python tests/bench_kernels.py bwd --reps 10 > $O/bwd_occ4.txt 2>&1
BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd_occ3.so timeout 300 python tests/bench_kernels.py bwd --reps 10 > $O/bwd_occ3.txt 2>&1
BPX_TILE_ORDER=0 timeout 300 python tests/bench_kernels.py wgrad --reps 10 > $O/wgrad_order0.txt 2>&1
BPX_TILE_ORDER=1 timeout 300 python tests/bench_kernels.py wgrad --reps 10 > $O/wgrad_order1.txt 2>&1
timeout 400 python bench.py --mode train --no-cpu-baseline > $O/train_fused.json 2> $O/train_fused.err
BPX_BWD_FUSED=0 timeout 400 python bench.py --mode train --no-cpu-baseline > $O/train_sep.json 2> $O/train_sep.err
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "bit_reproducible or cfg2_arch or network_against_reference or mix16" 2>&1 | tail -15 > $O/tests_net.txt
cat $O/tests_fused.txt $O/bwd_occ4.txt $O/bwd_occ3.txt; tail -3 $O/tests_net.txt
