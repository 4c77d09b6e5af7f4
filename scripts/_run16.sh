timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "split_k or network_cfg2 or reproducible or mixed_training_follows" 2>&1 | tail -5
for i in 1 2; do
for V in "BPX_SPLITK=0" "BPX_SPLITK_TARGET=256" "BPX_SPLITK_TARGET=512" "BPX_SPLITK_TARGET=1024" "BPX_SPLITK=0 BPX_WGRAD_CAP=50" "BPX_SPLITK=0 BPX_WGRAD_CAP=25"; do
  echo "== $V"
  env $V timeout 400 python bench.py --mode train --no-cpu-baseline --no-bf16-record --steps 30 2>&1 | grep "train record" | cut -c60-140
done; done
env BPX_SPLITK=0 timeout 300 python bench.py --mode train --breakdown --no-cpu-baseline 2>&1 | grep -E "16, 16, 16|, 8, 8, 8" > gpurun_out/bd_split0.txt
timeout 300 python bench.py --mode train --breakdown --no-cpu-baseline 2>&1 | grep -E "16, 16, 16|, 8, 8, 8" > gpurun_out/bd_split256.txt
env BPX_SPLITK_TARGET=512 timeout 300 python bench.py --mode train --breakdown --no-cpu-baseline 2>&1 | grep -E "16, 16, 16|, 8, 8, 8" > gpurun_out/bd_split512.txt
