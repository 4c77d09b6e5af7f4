#!/bin/bash
# Last call of round 6 on the final tree: the whole GPU suite, smoke(), the HBM-traffic counters (stamped with the sha256 of csrc/), the bench lines that
# quote them, kernel-trace statistics of the train / inference commands, the timeline of one graph-replayed step and the single-GPU checksums of the
# cfg-3 volumes.  Outputs: gpurun_out/r06_final/ (copy into profiles/ afterwards).
O=gpurun_out/r06_final; mkdir -p $O
export TMPDIR=/tmp; ROOT=$(pwd)
( time python -m pytest tests -m gpu -q -p no:cacheprovider ) > $O/r06_gpu_tests.txt 2>&1; tail -4 $O/r06_gpu_tests.txt
[ -f gpurun_out/diag_values.txt ] && cp gpurun_out/diag_values.txt $O/r06_gpu_test_values.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
cd /tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $ROOT/$O/pmc_f -o p -- python $ROOT/bench.py --mode train --steps 2 --warmup 1 --no-cpu-baseline --no-bf16-record --graph off > $ROOT/$O/pmc_f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $ROOT/$O/pmc_w -o p -- python $ROOT/bench.py --mode train --steps 2 --warmup 1 --no-cpu-baseline --no-bf16-record --graph off > $ROOT/$O/pmc_w.log 2>&1
cd $ROOT
python scripts/pmc_traffic.py $(find $O/pmc_f -name "p_results.db" | head -1) $(find $O/pmc_w -name "p_results.db" | head -1) > $O/pmc_traffic.json 2> $O/pmc_traffic.err
cp $O/pmc_traffic.json profiles/pmc_traffic.json
rm -rf $O/pmc_f $O/pmc_w
python bench.py --mode sliding --vol 1024 --steps 1 --warmup 1 --no-cpu-baseline > $O/r06_bench_sliding_1024.json 2> $O/bench_sliding.err
python bench.py --mode sliding --vol 512 --steps 2 --warmup 1 --no-cpu-baseline > $O/r06_bench_sliding_512.json 2>> $O/bench_sliding.err
python - "$O" r06 <<'PY' > $O/sliding_checksums.json
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
cp $O/sliding_checksums.json profiles/sliding_checksums.json
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench_driver_args.json 2> $O/bench_driver.err
python bench.py > $O/r06_bench.json 2> $O/bench.err
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$O/kt -o train -- python $ROOT/bench.py --mode train --steps 20 --warmup 3 --no-cpu-baseline --no-launch-events --no-bf16-record > $ROOT/$O/kt_train.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$O/kt -o infer -- python $ROOT/bench.py --mode infer --steps 20 --warmup 3 --no-cpu-baseline --no-launch-events > $ROOT/$O/kt_infer.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $ROOT/$O/kt -o t -- python $ROOT/bench.py --mode train --feed device --steps 6 --warmup 3 --no-cpu-baseline --no-launch-events --no-bf16-record > $ROOT/$O/kt_t.log 2>&1
cd $ROOT
cp $(find $O/kt -name "train_kernel_stats.csv" | head -1) $O/r06_bench_train_kernel_stats.csv
cp $(find $O/kt -name "infer_kernel_stats.csv" | head -1) $O/r06_bench_infer_kernel_stats.csv
python scripts/step_timeline.py $(find $O/kt -name "t_kernel_trace.csv" | head -1) > $O/r06_step_timeline.txt 2> $O/timeline.err
rm -rf $O/kt
for kg in 1 0 1 0; do BPX_CONV_KG=$kg python bench.py --mode train --feed device --steps 40 --warmup 8 --no-cpu-baseline --no-bf16-record --no-launch-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BPX_CONV_KG=$kg train ms_per_step %.4f (device-resident batch)' % d['ms_per_step'])"; done > $O/step_kg_final.txt
cat $O/step_kg_final.txt; tail -c 300 $O/r06_bench_driver_args.json; tail -3 $O/r06_step_timeline.txt
