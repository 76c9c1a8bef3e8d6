#!/bin/bash
timeout 120 python scratch/gmm_only.py
timeout 120 python scratch/bx_err.py 2>&1 | tail -3
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
