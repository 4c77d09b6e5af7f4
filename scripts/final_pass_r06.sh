#!/bin/bash
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused_conv_backward or role_split or resunet_train or smoke or reproduc" 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash scripts/pmc_redo_r06.sh 2>&1 | tail -5
( echo "== role-split form (conv3_bwd_rs_kernel, round 6)"; BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd_stamps.so python scripts/bwd_rs_stamps.py 128 48; BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd_stamps.so python scripts/bwd_rs_stamps.py 128 16
  echo "== serial form (conv3_bwd_kernel, BPX_BWD_RS=0)"; BPX_BWD_RS=0 BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd_stamps.so python scripts/bwd_stamps.py 128 16; BPX_BWD_RS=0 BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd_stamps.so python scripts/bwd_stamps.py 128 48 ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_final/r06_stamps_bwd_fused.txt
