#!/bin/bash
# The two PMC passes behind profiles/pmc_traffic.json alone (they have to follow the LAST change of a kernel source: the file carries the sha256
# of csrc/ and bench.py quotes `roofline.traffic` only when it matches the tree).  gpurun --timeout 900 -- 'bash scripts/refresh_pmc_traffic.sh'
O=gpurun_out/profiles_new
mkdir -p $O
export TMPDIR=/tmp
ROOT=$(pwd)
cd /tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $ROOT/$O/pmc_f -o p -- python $ROOT/bench.py --mode train --steps 2 --warmup 1 --no-cpu-baseline --no-bf16-record --graph off > $ROOT/$O/pmc_f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $ROOT/$O/pmc_w -o p -- python $ROOT/bench.py --mode train --steps 2 --warmup 1 --no-cpu-baseline --no-bf16-record --graph off > $ROOT/$O/pmc_w.log 2>&1
cd $ROOT
python scripts/pmc_traffic.py $(find $O/pmc_f -name "p_results.db" | head -1) $(find $O/pmc_w -name "p_results.db" | head -1) > $O/pmc_traffic.json 2> $O/pmc_traffic.err
rm -rf $O/pmc_f $O/pmc_w
cat $O/pmc_traffic.json; cat $O/pmc_traffic.err
