#!/bin/bash
for w in 5 60 5 200; do python bench.py --mode train --steps 20 --warmup $w --no-cpu-baseline --no-launch-events --no-bf16-record 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('warmup $w steps 20: ms_per_step %.4f host_fed(second region) %.4f' % (d['ms_per_step'], d.get('host_fed_ms_per_step')))"; done
