timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "dropout or network_cfg2_arch or reproducible or resunet_matches or graphed_train_step" 2>&1 | tail -15
