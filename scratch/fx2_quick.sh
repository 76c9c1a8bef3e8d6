#!/bin/bash
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python bench.py --steps 200 --warmup 20 2>&1 | tail -1 > gpurun_out/bench_fx2.json; cut -c1-300 gpurun_out/bench_fx2.json
