#!/bin/bash
timeout 900 python -m pytest tests -m gpu -x -q -k "ivector or iv_ or config4 or enroll" 2>&1 | tail -5
bash scratch/iv_prof.sh x 2>&1 | head -8
FB_IV_SOLVE=right bash scratch/iv_prof.sh x 2>&1 | head -3
timeout 300 python scratch/fuzz_iv.py 21 60 | tail -3
