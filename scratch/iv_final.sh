#!/bin/bash
R=$GRAFT_REPO_ROOT; tag=$1
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_iv_s1 -o p -- python $R/bench.py --arch iv --steps 40 --warmup 5 --streams 1 --no-cpu-baseline > $R/gpurun_out/${tag}_iv_bench_1attack.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_iv_s3 -o p -- python $R/bench.py --arch iv --steps 40 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
cd $R; python bench.py --arch iv > gpurun_out/${tag}_iv_bench.json 2>/dev/null
