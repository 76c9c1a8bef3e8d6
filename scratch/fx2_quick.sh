#!/bin/bash
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 300 python scratch/fuzz.py 131 200 | tail -2
FB_MFCC=r4 timeout 300 python scratch/fuzz.py 132 60 | tail -2
timeout 300 python scratch/fuzz_iv.py 133 60 | tail -2
python -c "import __graft_entry__ as g; g.smoke()"
timeout 300 python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_r16.json; cut -c1-200 gpurun_out/bench_r16.json
