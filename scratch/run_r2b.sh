cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2b
python -m pytest tests/test_gpu_fullsize_ivector.py tests/test_gpu_decisions.py -x -q -m gpu -s 2>&1 | tail -40 > gpurun_out/r2b/new_tests.log
python -m pytest tests -q -m gpu -s 2>&1 | grep "NDIFF\|passed\|failed" > gpurun_out/r2b/ndiff.log
