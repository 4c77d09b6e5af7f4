for i in 1 2; do
for L in tree ab; do
  if [ "$L" = ab ]; then export BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd_ab.so; else unset BPX_LIB_PATH; fi
  echo "== lib: $L"
  timeout 400 python bench.py --mode train --no-cpu-baseline --no-bf16-record --steps 30 2>&1 | grep "train record" | cut -c60-140
  [ $i = 1 ] && timeout 300 python tests/bench_kernels.py rcan 2>&1 | grep "graph replay: forward"
done; done
