cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2d
python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -30 > gpurun_out/r2d/par.log
