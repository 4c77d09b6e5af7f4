O=gpurun_out/r04c; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "fused_conv_backward" 2>&1 | tail -15 > $O/tests_fused.txt
timeout 300 python tests/bench_kernels.py bwd --reps 10 > $O/bwd.txt 2>&1
BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd_noldsw.so timeout 300 python tests/bench_kernels.py bwd --reps 10 > $O/bwd_noldsw.txt 2>&1
export BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd_stamps.so
timeout 200 python scripts/bwd_stamps.py 128 16 > $O/stamps_16.txt 2>&1
timeout 200 python scripts/bwd_stamps.py 128 48 > $O/stamps_48.txt 2>&1
unset BPX_LIB_PATH
timeout 400 python bench.py --mode train --no-cpu-baseline > $O/train_fused.json 2> $O/train_fused.err
cat $O/*.txt | grep -v amdgpu.ids; grep "train record" $O/train_fused.err
