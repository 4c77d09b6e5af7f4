O=gpurun_out/r04i; mkdir -p $O
timeout 400 python bench.py --mode train --no-cpu-baseline --no-bf16-record --steps 30 2>&1 | grep "train record"
BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd_ab.so timeout 400 python bench.py --mode train --no-cpu-baseline --no-bf16-record --steps 30 2>&1 | grep "train record\|Error\|error" | head -3
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "conv3d or fused_maxpool or cfg2_arch or network_against_reference or bit_reproducible or groupnorm or planar" 2>&1 | tail -4
