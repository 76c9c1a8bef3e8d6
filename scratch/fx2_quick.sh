#!/bin/bash
timeout 900 python -m pytest tests -m gpu -x -q -k "ivector or iv_ or config4 or enroll" 2>&1 | tail -3
timeout 300 python scratch/fuzz_iv.py 101 100 | tail -2
bash scratch/iv_prof.sh x 2>&1 | grep -E "solve|it/s"
