O=gpurun_out/r04b; mkdir -p $O
export BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd_stamps.so
timeout 200 python scripts/bwd_stamps.py 128 16 > $O/stamps_16.txt 2>&1
timeout 200 python scripts/bwd_stamps.py 128 48 > $O/stamps_48.txt 2>&1
unset BPX_LIB_PATH
timeout 300 python tests/bench_kernels.py bwd --reps 10 > $O/bwd_occ4.txt 2>&1
BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd_occ3.so timeout 300 python tests/bench_kernels.py bwd --reps 10 > $O/bwd_occ3.txt 2>&1
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "fused_conv_backward" 2>&1 | tail -5 > $O/tests_fused.txt
cat $O/*.txt | grep -v amdgpu.ids
