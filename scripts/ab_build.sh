#!/bin/bash
# Builds the kernel sources of a git revision as biapy_amd/libbiapy_amd_ab.so, next to the tree's library, for same-box A/B runs:
#   bash scripts/ab_build.sh HEAD~1
#   gpurun -- 'python tests/bench_kernels.py conv_fwd; BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd_ab.so python tests/bench_kernels.py conv_fwd'
# (box-to-box spread on the pool is +-5 % per kernel: only numbers from one call on one box compare)
set -e
REV=${1:-HEAD}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
git -C "$ROOT" archive "$REV" biapy_amd/csrc include | tar -x -C "$TMP"
make -C "$TMP/biapy_amd/csrc" -j8 OUT="$ROOT/biapy_amd/libbiapy_amd_ab.so" > "$TMP/build.log" 2>&1 || { tail -20 "$TMP/build.log"; exit 1; }
rm -rf "$TMP"
ls -la "$ROOT/biapy_amd/libbiapy_amd_ab.so"
