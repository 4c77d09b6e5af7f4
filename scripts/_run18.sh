timeout 300 python tests/bench_kernels.py prologue --reps 20 2>&1 | grep prologue
export BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd_bigtile.so
for O in 0 1 2; do
echo "== 4x8x16 tile (default), layer $O"; timeout 200 python tests/bench_kernels.py conv_fwd --only $O 2>&1 | grep conv_fwd
echo "== 4x16x16 tile, layer $O"; BPX_BIG_TILE=1 timeout 200 python tests/bench_kernels.py conv_fwd --only $O 2>&1 | grep conv_fwd
done
unset BPX_LIB_PATH
for V in "BPX_WGRAD_CAP=100" "BPX_WGRAD_CAP=200" "BPX_WGRAD_CAP=400" "BPX_WGRAD_CAP=100" "BPX_WGRAD_CAP=200" "BPX_WGRAD_CAP=400"; do
  echo "== $V"
  env $V timeout 400 python bench.py --mode train --no-cpu-baseline --no-bf16-record --steps 30 2>&1 | grep "train record" | cut -c60-140
done
