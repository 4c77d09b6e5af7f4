O=gpurun_out/r04h; mkdir -p $O
for i in 11 12 13 14 15 16; do timeout 100 python tests/bench_kernels.py conv_fwd --only $i --reps 20 2>&1 | grep conv_fwd; timeout 100 python tests/bench_kernels.py conv_dgrad --only $i --reps 20 2>&1 | grep conv_dgrad; done > $O/new.txt
export BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd_ab.so
for i in 11 12 13 14 15 16; do timeout 100 python tests/bench_kernels.py conv_fwd --only $i --reps 20 2>&1 | grep conv_fwd; timeout 100 python tests/bench_kernels.py conv_dgrad --only $i --reps 20 2>&1 | grep conv_dgrad; done > $O/old.txt
unset BPX_LIB_PATH
paste -d'\n' $O/old.txt $O/new.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "conv3d_forward or conv3d_backward or kernel_variants or cfg2_arch" 2>&1 | tail -3
timeout 400 python bench.py --mode train --no-cpu-baseline --no-bf16-record --steps 30 2>&1 | grep "train record"
