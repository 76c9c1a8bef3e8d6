#!/bin/bash
for b in 1 0; do echo "FB_GMM_BALANCE=$b"; FB_GMM_BALANCE=$b timeout 120 python scratch/gmm_only.py; done
timeout 120 python scratch/bx_err.py 2>&1 | tail -1
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python scratch/fuzz.py 91 150 | tail -3
timeout 300 python scratch/fuzz_iv.py 92 40 | tail -2
for b in 1 0; do FB_GMM_BALANCE=$b timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-160; done
