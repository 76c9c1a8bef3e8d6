#!/bin/bash
for k in 2 3 4 5 6; do echo -n "streams $k: "; python bench.py --steps 150 --warmup 20 --no-cpu-baseline --streams $k 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['ms_per_step'])"; done
