R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for tag in fuse nofuse; do
  if [ $tag = nofuse ]; then export FB_NO_FUSE=1; else unset FB_NO_FUSE; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2e_${tag}_s1 -o p -- python $R/bench.py --steps 100 --warmup 10 --streams 1 --no-cpu-baseline --precondition 10 > $R/gpurun_out/r2e_${tag}_s1_bench.json 2>/dev/null
done
cd $R
for tag in fuse nofuse; do f=$(find gpurun_out/r2e_${tag}_s1 -name "*kernel_stats.csv" | head -1); echo "== $tag $f"; head -14 $f | cut -d, -f1-8; done
