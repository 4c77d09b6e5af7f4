#!/bin/bash
O=$PWD/gpurun_out/r06_c15; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pool or resunet or unet" 2>&1 | tail -2
export TMPDIR=/tmp; ROOT=$PWD; cd /tmp
for L in "" _ab; do
BPX_LIB_PATH=$ROOT/biapy_amd/libbiapy_amd$L.so timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt$L -o train -- python $ROOT/bench.py --mode train --steps 20 --warmup 3 --no-cpu-baseline --no-launch-events --no-bf16-record > $O/kt$L.log 2>&1
cp $(find $O/kt$L -name "train_kernel_stats.csv" | head -1) $O/train_kernel_stats$L.csv; rm -rf $O/kt$L
echo "== lib$L"; grep -E "maxpool_bwd|conv3_bwd|conv3_lp_kernel<4, 4, 16, 2, 0" $O/train_kernel_stats$L.csv | cut -c1-140
done
