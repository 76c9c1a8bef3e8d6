cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2c
python -m pytest tests -q -m gpu -x 2>&1 | tail -5 > gpurun_out/r2c/all.log
python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r2c/bench.json 2> gpurun_out/r2c/bench.err
FB_NO_FUSE=1 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r2c/bench_nofuse.json 2>> gpurun_out/r2c/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2c/bench20.json 2>> gpurun_out/r2c/bench.err
