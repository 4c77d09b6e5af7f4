#!/bin/bash
# HBM-traffic counters again (the role-split kernel's name was missing from the entry-point map of scripts/pmc_traffic.py), then the bench lines that quote them
O=gpurun_out/r06_final; mkdir -p $O
export TMPDIR=/tmp; ROOT=$(pwd); cd /tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $ROOT/$O/pmc_f -o p -- python $ROOT/bench.py --mode train --steps 2 --warmup 1 --no-cpu-baseline --no-bf16-record --graph off > $ROOT/$O/pmc_f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $ROOT/$O/pmc_w -o p -- python $ROOT/bench.py --mode train --steps 2 --warmup 1 --no-cpu-baseline --no-bf16-record --graph off > $ROOT/$O/pmc_w.log 2>&1
cd $ROOT
python scripts/pmc_traffic.py $(find $O/pmc_f -name "p_results.db" | head -1) $(find $O/pmc_w -name "p_results.db" | head -1) > $O/pmc_traffic.json 2> $O/pmc_traffic.err
cp $O/pmc_traffic.json profiles/pmc_traffic.json
rm -rf $O/pmc_f $O/pmc_w
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench_driver_args.json 2> $O/bench_driver.err
python bench.py > $O/r06_bench.json 2> $O/bench.err
python -c "
import json
d=json.load(open('$O/pmc_traffic.json'))
for k,v in d.items():
    if k!='_meta': print(k, v['launches'], round(v['total_bytes']/1e6,1),'MB per call')
"
