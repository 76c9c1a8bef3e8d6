#!/bin/bash
for i in 1 2; do timeout 120 python scratch/gmm_only.py; done
timeout 120 python scratch/bx_err.py 2>&1 | tail -3
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
timeout 300 python scratch/fuzz.py 61 150 | tail -3
