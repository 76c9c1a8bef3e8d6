#!/bin/bash
timeout 900 python -m pytest tests -m gpu -x -q -k "ivector or iv_ or config4 or enroll or fuzz" 2>&1 | tail -3
timeout 300 python scratch/fuzz_iv.py 111 120 | tail -2
bash scratch/iv_prof.sh x 2>&1 | grep -E "stats|contract|it/s"
bash scratch/pmc_iv.sh pf2 WRITE_SIZE 2>&1 | grep stats
timeout 300 python bench.py --arch iv --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-150
