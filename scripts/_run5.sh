O=gpurun_out/r04e; mkdir -p $O
timeout 1500 python tests/gpu_diag.py --net --out $O/r04_gpu_diag.txt > $O/diag.log 2>&1
( time timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -s -k "mixed_training_follows" ) > $O/loss_curve.txt 2>&1
grep -c "^ok" $O/r04_gpu_diag.txt; grep "FAIL\|level\|checks passed" $O/r04_gpu_diag.txt | head -60; tail -8 $O/loss_curve.txt
