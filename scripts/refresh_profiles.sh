#!/bin/bash
# Regenerates every file of profiles/ in ONE run on one GPU box (run through gpurun from the repo root):
#   bash scripts/ab_build_flags.sh zmstamps -DBPX_ZM_STAMPS -DBPX_ZM_SPLIT=8 ; bash scripts/ab_build_flags.sh stamps -DBPX_BWD_STAMPS      (profiling builds, before the call)
#   gpurun --timeout 3600 -- 'bash scripts/refresh_profiles.sh r06'
# Outputs land in gpurun_out/profiles_new/ (merged back by gpurun); copy them into profiles/ afterwards.
# PMC counters are collected in their own passes with --kernel-trace only (no sys/hip/hsa trace domains).
R=${1:-r06}
O=gpurun_out/profiles_new
mkdir -p $O
export TMPDIR=/tmp
ROOT=$(pwd)
# HBM traffic first: bench.py reads profiles/pmc_traffic*.json (stamped with the sha256 of csrc/) for `roofline.traffic`, so the counters of THIS tree
# must be in place before the bench line is produced
cd /tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $ROOT/$O/pmc_f -o p -- python $ROOT/bench.py --mode train --steps 2 --warmup 1 --no-cpu-baseline --no-bf16-record --graph off > $ROOT/$O/pmc_f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $ROOT/$O/pmc_w -o p -- python $ROOT/bench.py --mode train --steps 2 --warmup 1 --no-cpu-baseline --no-bf16-record --graph off > $ROOT/$O/pmc_w.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $ROOT/$O/pmc_tf -o p -- python $ROOT/tests/bench_kernels.py merge_rows --reps 4 > $ROOT/$O/pmc_tf.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $ROOT/$O/pmc_tw -o p -- python $ROOT/tests/bench_kernels.py merge_rows --reps 4 > $ROOT/$O/pmc_tw.log 2>&1
cd $ROOT
python scripts/pmc_traffic.py $(find $O/pmc_f -name "p_results.db" | head -1) $(find $O/pmc_w -name "p_results.db" | head -1) > $O/pmc_traffic.json 2> $O/pmc_traffic.err
python scripts/pmc_traffic.py $(find $O/pmc_tf -name "p_results.db" | head -1) $(find $O/pmc_tw -name "p_results.db" | head -1) > $O/pmc_traffic_tiling.json 2>> $O/pmc_traffic.err
cp $O/pmc_traffic.json profiles/pmc_traffic.json; cp $O/pmc_traffic_tiling.json profiles/pmc_traffic_tiling.json
( time python -m pytest tests -m gpu -q -p no:cacheprovider ) > $O/${R}_gpu_tests.txt 2>&1
python bench.py > $O/${R}_bench.json 2> $O/bench.err
python bench.py --force-ddp --self-check --mode train --no-cpu-baseline > $O/${R}_bench_train_dp_1rank.json 2> $O/bench_dp.err
python bench.py --mode sliding --vol 1024 --steps 1 --warmup 1 --no-cpu-baseline > $O/${R}_bench_sliding_1024.json 2> $O/bench_sliding.err
python bench.py --mode sliding --vol 512 --steps 2 --warmup 1 --no-cpu-baseline > $O/${R}_bench_sliding_512.json 2>> $O/bench_sliding.err
# the single-GPU checksums of the blended cfg-3 volumes (initial weights of seed 0): what `bench.py --gpus N` compares its gathered volume with
python - "$O" "$R" <<'PY' > $O/sliding_checksums.json
import json, sys
O, R = sys.argv[1:3]
out = {}
for vol, path, pick in ((512, f"{O}/{R}_bench_sliding_512.json", lambda d: d.get("sliding") or d), (1024, f"{O}/{R}_bench_sliding_1024.json", lambda d: d.get("sliding") or d)):
    try:
        rec = pick(json.loads(open(path).read().strip().splitlines()[-1]))
        out[str(vol)] = dict(checksum=rec["checksum"], dtype=rec.get("dtype"), patches=rec.get("config", {}).get("patches"), source=path.split("/")[-1])
    except Exception as e:  # noqa: BLE001
        print(f"sliding_checksums: {vol}: {e}", file=sys.stderr)
print(json.dumps(out, indent=1))
PY
BPX_BENCH_ONE_DEVICE=1 BPX_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 \
   bench.py --gpus 2 --steps 5 --warmup 2 --vol 512 > $O/${R}_bench_2ranks_one_gpu_gloo.json 2> $O/bench_2rank.err
python bench.py --arch resunetpp --batch 4 --steps 5 --warmup 2 > $O/${R}_bench_resunetpp_80.json 2> $O/bench_pp.err
python bench.py --arch resunetpp --batch 4 --breakdown --graph off > $O/${R}_breakdown_resunetpp_events.txt 2> /dev/null
python bench.py --breakdown --graph off --mode train > $O/${R}_breakdown_train_events.txt 2> /dev/null
python bench.py --breakdown --graph off --mode infer > $O/${R}_breakdown_infer_events.txt 2> /dev/null
python tests/bench_kernels.py merge > $O/${R}_merge_crop.txt 2>&1
( python scripts/dgrad_stamps.py 64 96 32; python scripts/dgrad_stamps.py 64 32 32; python scripts/conv_stamps.py 0; python scripts/conv_stamps.py 11; python scripts/hbm_rw_probe.py ) 2>&1 | grep -v amdgpu.ids > $O/${R}_stamps_fwd_dgrad.txt
# fused backward: per-phase cycle stamps need the profiling build (bash scripts/ab_build_flags.sh stamps -DBPX_BWD_STAMPS, done before the gpurun call)
if [ -f biapy_amd/libbiapy_amd_stamps.so ]; then
  ( echo "== role-split form (conv3_bwd_rs_kernel, round 6)"; BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd_stamps.so python scripts/bwd_rs_stamps.py 128 48; BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd_stamps.so python scripts/bwd_rs_stamps.py 128 16
    echo "== serial form (conv3_bwd_kernel, BPX_BWD_RS=0)"; BPX_BWD_RS=0 BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd_stamps.so python scripts/bwd_stamps.py 128 16; BPX_BWD_RS=0 BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd_stamps.so python scripts/bwd_stamps.py 128 48 ) 2>&1 | grep -v amdgpu.ids > $O/${R}_stamps_bwd_fused.txt
fi
( for rep in 1 2; do echo "== role-split (default)"; python tests/bench_kernels.py bwd --reps 10; echo "== serial fused kernel (BPX_BWD_RS=0)"; BPX_BWD_RS=0 python tests/bench_kernels.py bwd --reps 10; done ) 2>&1 | grep -v amdgpu.ids > $O/${R}_bwd_fused_vs_separate.txt
for rs in 3 0 3 0; do BPX_BWD_RS=$rs python bench.py --mode train --feed device --steps 40 --warmup 8 --no-cpu-baseline --no-bf16-record --no-launch-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BPX_BWD_RS=$rs train ms_per_step %.4f (device-resident batch)' % d['ms_per_step'])"; done > $O/${R}_step_bwd_rs_ab.txt 2>&1
# round 5: the z-marching forward kernel of the level-0 layers against the lean kernel (same call), its per-phase cycle stamps, its runtime occupancy
( python -c "
from biapy_amd import _lib as L
print('resident workgroups per CU as the runtime computes them: conv3_zm_kernel<1> (one input chunk)', L.lib.bpx_debug_conv_zm_occupancy(1), ', <3> (three chunks)', L.lib.bpx_debug_conv_zm_occupancy(3), ', conv3_zs_kernel (role split, 512 threads)', L.lib.bpx_debug_conv_zm_occupancy(0))"
  for rep in 1 2; do
    echo "== z-march"; BPX_CONV_ZM=1 python tests/bench_kernels.py conv_fwd --dtype f16 --only 0,1,2 --reps 20 | grep conv_fwd
    echo "== lean (BPX_CONV_ZM=0)"; BPX_CONV_ZM=0 python tests/bench_kernels.py conv_fwd --dtype f16 --only 0,1,2 --reps 20 | grep conv_fwd
    echo "== z-march without the role split (BPX_CONV_ZS=0: conv3_zm_kernel<1> instead of conv3_zs_kernel)"; BPX_CONV_ZS=0 python tests/bench_kernels.py conv_fwd --dtype f16 --only 2 --reps 20 | grep conv_fwd
  done ) 2>&1 | grep -v amdgpu.ids > $O/${R}_zmarch_vs_lean.txt
if [ -f biapy_amd/libbiapy_amd_zmstamps.so ]; then
  ( for k in 2 1 0; do BPX_STAMP_SPLIT=1 BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd_zmstamps.so BPX_CONV_ZM=6 python scripts/zm_stamps.py $k; done ) 2>&1 | grep -v amdgpu.ids > $O/${R}_stamps_zmarch.txt
fi
for zm in 1 0; do BPX_CONV_ZM=$zm python bench.py --mode train --steps 40 --warmup 8 --no-cpu-baseline --no-bf16-record --no-launch-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BPX_CONV_ZM=$zm train ms_per_step %.4f' % d['ms_per_step'])"; done > $O/${R}_step_zmarch_ab.txt 2>&1
for zm in 1 0; do BPX_CONV_ZM=$zm python bench.py --mode infer --steps 40 --warmup 8 --no-cpu-baseline --no-launch-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BPX_CONV_ZM=$zm infer ms_per_step %.4f' % d['ms_per_step'])"; done >> $O/${R}_step_zmarch_ab.txt 2>&1
# round 6: two K groups in the small-tile kernel of the <= 16^3 layers (conv3_kernel<..., KG = 2>): kernel times and the whole step, alternating
( echo "# two K groups in the small-tile conv kernel (conv3_kernel<..., KG = 2>), kernel times by HIP events (tests/bench_kernels.py conv_fwd), same box, alternating"
  for kg in 1 0 1 0; do echo "== BPX_CONV_KG=$kg"; BPX_CONV_KG=$kg python tests/bench_kernels.py conv_fwd 2>&1 | grep -E "^conv_fwd +(16|8)\^3"; done
  echo "# whole step, 40 graph-replayed steps, same box, alternating (bench.py --mode train / infer)"
  for kg in 1 0 1 0; do BPX_CONV_KG=$kg python bench.py --mode train --feed device --steps 40 --warmup 8 --no-cpu-baseline --no-bf16-record --no-launch-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BPX_CONV_KG=$kg train ms_per_step %.4f (device-resident batch)' % d['ms_per_step'])"; done
  for kg in 1 0 1 0; do BPX_CONV_KG=$kg python bench.py --mode infer --steps 40 --warmup 8 --no-cpu-baseline --no-launch-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BPX_CONV_KG=$kg infer ms_per_step %.4f' % d['ms_per_step'])"; done ) 2>&1 | grep -v amdgpu.ids > $O/${R}_step_conv_kg_ab.txt
( python tests/bench_kernels.py k1 --reps 20; python tests/bench_kernels.py pws --reps 20 ) 2>&1 | grep -v amdgpu.ids > $O/${R}_stream_vs_tile.txt
# round 5: transposed-conv forward of level 0 (convt_k1_kernel against pw_kernel) and the first layer (buffer- against pointer-addressed), event timing
# in the network; the store-pattern probe
( for rep in 1 2; do
    for k1 in 1 0; do echo "== BPX_CONVT_K1=$k1"; BPX_CONVT_K1=$k1 python bench.py --breakdown --graph off --mode infer 2>/dev/null | grep "convT3d_k2s2_fwd (2, 4, 64\|c1_fwd"; done
    echo "== first layer, pointer-addressed instance (BPX_C1_PERSIST bit 30)"; BPX_C1_PERSIST=$((2048 + (1 << 30))) python bench.py --breakdown --graph off --mode infer 2>/dev/null | grep "c1_fwd"
  done
  echo "== store patterns (scripts/probes/store_pattern_probe.hip)"; [ -x scripts/probes/store_pattern_probe.bin ] && scripts/probes/store_pattern_probe.bin ) 2>&1 | grep -v amdgpu.ids > $O/${R}_convt_c1_ab.txt
python tests/gpu_diag.py --net --out $O/${R}_gpu_diag.txt > /dev/null 2>&1
[ -f gpurun_out/diag_values.txt ] && cp gpurun_out/diag_values.txt $O/${R}_gpu_test_values.txt     # measured values the parity tests recorded (loss-curve gap, Dice rows)
python tests/bench_kernels.py rcan 2>&1 | grep -v "Warning\|run_backward" > $O/${R}_rcan_trunk_64.txt
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$O/kt -o train -- python $ROOT/bench.py --mode train --steps 20 --warmup 3 --no-cpu-baseline --no-launch-events --no-bf16-record > $ROOT/$O/kt_train.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$O/kt -o resunetpp -- python $ROOT/bench.py --arch resunetpp --batch 4 --steps 20 --warmup 3 > $ROOT/$O/kt_pp.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$O/kt -o infer -- python $ROOT/bench.py --mode infer --steps 20 --warmup 3 --no-cpu-baseline --no-launch-events > $ROOT/$O/kt_infer.log 2>&1
cd $ROOT
cp $(find $O/kt -name "train_kernel_stats.csv" | head -1) $O/${R}_bench_train_kernel_stats.csv
cp $(find $O/kt -name "infer_kernel_stats.csv" | head -1) $O/${R}_bench_infer_kernel_stats.csv
cp $(find $O/kt -name "resunetpp_kernel_stats.csv" | head -1) $O/${R}_bench_resunetpp_kernel_stats.csv
cd /tmp
timeout 400 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES --kernel-trace \
   -d $ROOT/$O/pmc_a -o p -- python $ROOT/bench.py --mode train --steps 2 --warmup 1 --no-cpu-baseline --no-bf16-record --graph off > $ROOT/$O/pmc_a.log 2>&1
timeout 400 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace \
   -d $ROOT/$O/pmc_b -o p -- python $ROOT/bench.py --mode train --steps 2 --warmup 1 --no-cpu-baseline --no-bf16-record --graph off > $ROOT/$O/pmc_b.log 2>&1
cd $ROOT
for k in conv3_bwd_rs_kernel conv3_zm_kernel conv3_zs_kernel conv3_lp_kernel wgrad_sdm_kernel conv3_bwd_kernel pw_nbs_kernel wgrad_k1_dma_kernel; do
  python scripts/pmc_report.py $(find $O/pmc_a -name "p_results.db" | head -1) $k
  python scripts/pmc_report.py $(find $O/pmc_b -name "p_results.db" | head -1) $k
done > $O/${R}_pmc_sq_conv_wgrad.txt 2>&1
rm -rf $O/kt $O/pmc_f $O/pmc_w $O/pmc_a $O/pmc_b $O/pmc_tf $O/pmc_tw
ls -la $O
tail -3 $O/${R}_gpu_tests.txt; tail -c 400 $O/${R}_bench.json; cat $O/pmc_traffic.json | head -c 700
