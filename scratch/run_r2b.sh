cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2b
python -m pytest tests/test_gpu_driver.py -x -q -m gpu 2>&1 | tail -30 > gpurun_out/r2b/driver.log
python -m pytest tests -q -m gpu 2>&1 | tail -5 > gpurun_out/r2b/all.log
