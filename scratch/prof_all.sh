#!/bin/bash
# kernel-trace stats (1 attack in flight and default 3) + HBM traffic PMC passes; outputs under gpurun_out/
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
tag=$1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_s1 -o p -- python $R/bench.py --steps 100 --warmup 10 --streams 1 --no-cpu-baseline > $R/gpurun_out/${tag}_s1_bench.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_s3 -o p -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $R/gpurun_out/${tag}_s3_bench.json 2>/dev/null
cd $R && scratch/pmc_traffic.sh > gpurun_out/${tag}_traffic.txt 2>&1
python bench.py > gpurun_out/${tag}_bench_default.json 2>/dev/null
tail -3 gpurun_out/${tag}_traffic.txt
cat gpurun_out/${tag}_bench_default.json | cut -c1-400
