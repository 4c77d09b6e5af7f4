#!/bin/bash
# Last GPU call of a round after a late kernel-source change: the PMC passes (stamp), then the bench line (with `roofline.traffic` from the fresh file),
# the RCAN trunk timings and the whole GPU suite on the final tree.  gpurun --timeout 570 -- 'bash scripts/final_pass.sh r04'
R=${1:-r04}
O=gpurun_out/profiles_new
mkdir -p $O
bash scripts/refresh_pmc_traffic.sh > $O/pmc_refresh.log 2>&1
if python -c "import json,sys; d=json.load(open('$O/pmc_traffic.json')); sys.exit(0 if d.get('_meta') else 1)"; then cp $O/pmc_traffic.json profiles/pmc_traffic.json; fi
python bench.py > $O/${R}_bench.json 2> $O/bench.err
python tests/bench_kernels.py rcan 2>&1 | grep -v "Warning\|run_backward\|amdgpu.ids" > $O/${R}_rcan_trunk_64.txt
( time python -m pytest tests -m gpu -q -p no:cacheprovider ) > $O/${R}_gpu_tests.txt 2>&1
tail -3 $O/${R}_gpu_tests.txt; tail -c 300 $O/${R}_bench.json; head -c 400 $O/pmc_traffic.json
