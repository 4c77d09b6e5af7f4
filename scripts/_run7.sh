timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "saturate" 2>&1 | grep -v "^$" | tail -30
