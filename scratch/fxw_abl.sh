cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2d
: > gpurun_out/r2d/abl.log
for v in 0; do
  if [ $v = 0 ]; then export FB_EXTRA_HIPCC_FLAGS=""; else export FB_EXTRA_HIPCC_FLAGS="-DFB_FXW_ABL=$v"; fi
  touch fakebob_amd/csrc/gmm_kernels.hip
  python -m fakebob_amd.build > gpurun_out/r2d/build_$v.log 2>&1
  ABL=$v python scratch/fxw_abl.py > gpurun_out/r2d/run_$v.log 2>&1
  grep ABL gpurun_out/r2d/run_$v.log >> gpurun_out/r2d/abl.log
done
FB_GMM_NARROW=1 ABL=narrow python scratch/fxw_abl.py 2>&1 | grep ABL >> gpurun_out/r2d/abl.log
cat gpurun_out/r2d/abl.log
python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_range_guard.py tests/test_gpu_fuzz.py tests/test_gpu_decisions.py -q -m gpu -x 2>&1 | tail -3
