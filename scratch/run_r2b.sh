cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2b
python -m pytest tests/test_gpu_range_guard.py -x -q -m gpu -s 2>&1 | tail -30 > gpurun_out/r2b/range.log
python -m pytest tests -q -m gpu 2>&1 | tail -5 > gpurun_out/r2b/all.log
python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r2b/bench.json 2> gpurun_out/r2b/bench.err
