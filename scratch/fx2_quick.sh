#!/bin/bash
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python scratch/fuzz_iv.py 81 120 | tail -3
bash scratch/iv_prof.sh x 2>&1 | grep -E "stats|blocks|count|fill|it/s"
timeout 300 python bench.py --arch iv --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-150
