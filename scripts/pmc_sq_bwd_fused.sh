#!/bin/bash
# SQ counters of the fused backward kernels on the bench shapes: serial (BPX_BWD_RS=0) and role-split (3)
O=$PWD/gpurun_out/r06_pmc; mkdir -p $O
export TMPDIR=/tmp; ROOT=$PWD; cd /tmp
for m in 0 3; do
  BPX_BWD_RS=$m timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES --kernel-trace -d $O/a$m -o p -- python $ROOT/tests/bench_kernels.py bwd --reps 3 > $O/a$m.log 2>&1
  BPX_BWD_RS=$m timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace -d $O/b$m -o p -- python $ROOT/tests/bench_kernels.py bwd --reps 3 > $O/b$m.log 2>&1
  BPX_BWD_RS=$m timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_IFETCH --kernel-trace -d $O/c$m -o p -- python $ROOT/tests/bench_kernels.py bwd --reps 3 > $O/c$m.log 2>&1
  for x in a b c; do python $ROOT/scripts/pmc_report.py $(find $O/$x$m -name "p_results.db" | head -1) conv3_bwd 2>&1 | grep -v "^kernel\|^void.*[0-9]%"; done > $O/rs$m.txt
  rm -rf $O/a$m $O/b$m $O/c$m
done
cat $O/rs0.txt; echo ======; cat $O/rs3.txt
