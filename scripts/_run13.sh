for L in "" nosat noext neither; do
  if [ -n "$L" ]; then export BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd_$L.so; fi
  echo "== lib: ${L:-tree}"
  timeout 300 python tests/bench_kernels.py rcan 2>&1 | grep "graph replay: forward"
done
