#!/bin/bash
# Regenerates every file of profiles/ in ONE run on one GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash scripts/refresh_profiles.sh r01'
# Outputs land in gpurun_out/profiles_new/ (merged back by gpurun); copy them into profiles/ afterwards.
# PMC counters are collected in their own passes with --kernel-trace only (no sys/hip/hsa trace domains).
R=${1:-r01}
O=gpurun_out/profiles_new
mkdir -p $O
export TMPDIR=/tmp
python bench.py > $O/${R}_bench_train.json 2> $O/bench_train.err
python bench.py --mode infer --no-cpu-baseline > $O/${R}_bench_infer.json 2> $O/bench_infer.err
python bench.py --force-ddp --no-cpu-baseline > $O/${R}_bench_train_dp_1rank.json 2> $O/bench_dp.err
python bench.py --mode sliding --vol 512 --steps 3 --warmup 1 --no-cpu-baseline > $O/${R}_bench_sliding_512.json 2> $O/bench_sliding.err
python bench.py --breakdown --graph off > $O/${R}_breakdown_train_events.txt 2> /dev/null
python bench.py --breakdown --graph off --mode infer > $O/${R}_breakdown_infer_events.txt 2> /dev/null
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o train -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/kt_train.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o infer -- python bench.py --mode infer --steps 10 --warmup 3 --no-cpu-baseline > $O/kt_infer.log 2>&1
cp $(find $O/kt -name "train_kernel_stats.csv" | head -1) $O/${R}_bench_train_kernel_stats.csv
cp $(find $O/kt -name "infer_kernel_stats.csv" | head -1) $O/${R}_bench_infer_kernel_stats.csv
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_f -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --graph off > $O/pmc_f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_w -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --graph off > $O/pmc_w.log 2>&1
python scripts/pmc_traffic.py $(find $O/pmc_f -name "p_results.db" | head -1) $(find $O/pmc_w -name "p_results.db" | head -1) > $O/pmc_traffic.json 2> $O/pmc_traffic.err
rm -rf $O/kt $O/pmc_f $O/pmc_w
ls -la $O
tail -c 600 $O/${R}_bench_train.json; cat $O/pmc_traffic.json | head -c 600
