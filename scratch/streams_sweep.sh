#!/bin/bash
for q in "" 8; do
for k in 2 3 4 5 6; do echo -n "GPU_MAX_HW_QUEUES=$q streams $k: "; GPU_MAX_HW_QUEUES=$q python bench.py --steps 150 --warmup 20 --no-cpu-baseline --streams $k 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['ms_per_step'])"; done
done
