R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_gsel2; mkdir -p $O; cd $R
bash tools/profile/gsel_instrumented.sh gpurun_out/r06_gsel2/gselw 2>&1 | tail -8
rm -f gpurun_out/r06_gsel2/gselw/*.o gpurun_out/r06_gsel2/gselw/*.so
