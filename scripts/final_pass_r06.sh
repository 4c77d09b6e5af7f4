#!/bin/bash
# the bench line on the final tree (default arguments and the driver's), the role-split stamps with the final script, the tests touched after the refresh
O=gpurun_out/r06_final   # (copied into profiles/ afterwards); mkdir -p $O
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/r06_bench_driver_args.json 2> $O/bench_driver.err
python bench.py > $O/r06_bench.json 2> $O/bench.err
( echo "== role-split form (conv3_bwd_rs_kernel, round 6)"; BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd_stamps.so python scripts/bwd_rs_stamps.py 128 48; BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd_stamps.so python scripts/bwd_rs_stamps.py 128 16
  echo "== serial form (conv3_bwd_kernel, BPX_BWD_RS=0)"; BPX_BWD_RS=0 BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd_stamps.so python scripts/bwd_stamps.py 128 16; BPX_BWD_RS=0 BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd_stamps.so python scripts/bwd_stamps.py 128 48 ) 2>&1 | grep -v amdgpu.ids > $O/r06_stamps_bwd_fused.txt
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "segmentation_losses or fused_conv_backward or role_split" 2>&1 | tail -2 > $O/tests_touched.txt
tail -3 $O/bench_driver.err; tail -c 300 $O/r06_bench.json; cat $O/tests_touched.txt
