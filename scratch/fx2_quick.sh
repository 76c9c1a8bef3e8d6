#!/bin/bash
timeout 900 python -m pytest tests -m gpu -x -q -k "ivector or iv_ or config4 or enroll" 2>&1 | tail -3
timeout 300 python scratch/fuzz_iv.py 71 100 | tail -2
FB_IV_SOLVE=dense timeout 300 python scratch/fuzz_iv.py 72 40 | tail -2
FB_IV_CONTRACT=reg timeout 300 python scratch/fuzz_iv.py 73 40 | tail -2
timeout 300 python bench.py --arch iv --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-150
