#!/bin/bash
# Round-2 GPU call A: the whole -m gpu suite, the crop/merge A-B, the new all-section bench line, the N>1 launch line on one
# GPU (2 ranks, gloo), and the SQ counter passes of one eager train step.
export TMPDIR=/tmp
O=gpurun_out/r2a
mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 --tb=short -p no:cacheprovider ) > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
python tests/bench_kernels.py merge > $O/merge_crop.txt 2>&1
cat $O/merge_crop.txt
( time python bench.py ) > $O/bench_all.json 2> $O/bench_all.err
tail -c 1500 $O/bench_all.json; tail -5 $O/bench_all.err
BPX_BENCH_ONE_DEVICE=1 BPX_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 \
   bench.py --gpus 2 --steps 3 --warmup 1 --vol 256 > $O/bench_2rank_gloo.json 2> $O/bench_2rank.err
tail -c 800 $O/bench_2rank_gloo.json; tail -3 $O/bench_2rank.err
cd /tmp
timeout 500 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES --kernel-trace \
   -d $GRAFT_REPO_ROOT/$O/pmc_a -o p -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 2 --warmup 1 --no-cpu-baseline --graph off > $GRAFT_REPO_ROOT/$O/pmc_a.log 2>&1
timeout 500 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace \
   -d $GRAFT_REPO_ROOT/$O/pmc_b -o p -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 2 --warmup 1 --no-cpu-baseline --graph off > $GRAFT_REPO_ROOT/$O/pmc_b.log 2>&1
cd $GRAFT_REPO_ROOT
for k in conv3_lp_kernel wgrad_sdm_kernel; do
  python scripts/pmc_report.py $(find $O/pmc_a -name "p_results.db" | head -1) $k
  python scripts/pmc_report.py $(find $O/pmc_b -name "p_results.db" | head -1) $k
done > $O/pmc_sq_conv_wgrad.txt 2>&1
rm -rf $O/pmc_a $O/pmc_b
head -c 3000 $O/pmc_sq_conv_wgrad.txt
