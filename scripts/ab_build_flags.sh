#!/bin/bash
# Builds the TREE's kernel sources with extra compiler flags as biapy_amd/libbiapy_amd_<tag>.so, next to the tree's library, for same-box A/B runs
# of compile-time switches (BPX_LIB_PATH selects the library):
#   bash scripts/ab_build_flags.sh occ3 -DBPX_BWD_OCC1=3
#   gpurun -- 'python tests/bench_kernels.py bwd; BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd_occ3.so python tests/bench_kernels.py bwd'
set -e
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
mkdir -p "$TMP/biapy_amd" "$TMP/include"
cp -r "$ROOT/biapy_amd/csrc" "$TMP/biapy_amd/csrc"
cp "$ROOT/include/"*.h "$TMP/include/"
rm -f "$TMP/biapy_amd/csrc/"*.o
make -C "$TMP/biapy_amd/csrc" -j8 OUT="$ROOT/biapy_amd/libbiapy_amd_$TAG.so" CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result $*" > "$TMP/build.log" 2>&1 || { tail -20 "$TMP/build.log"; exit 1; }
rm -rf "$TMP"
ls -la "$ROOT/biapy_amd/libbiapy_amd_$TAG.so"
