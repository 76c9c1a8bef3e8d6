cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2c
for k in 2 3 4 5 6; do
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-single --streams $k > gpurun_out/r2c/k_fuse_$k.json 2>/dev/null
FB_NO_FUSE=1 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-single --streams $k > gpurun_out/r2c/k_nofuse_$k.json 2>/dev/null
done
