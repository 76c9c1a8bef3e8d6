#!/bin/bash
# round 5, first GPU call: the -m gpu suite, the driver's bench line, and kernel-trace baselines of the i-vector SV chain
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_0; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
cd /tmp && export TMPDIR=/tmp
prof() {
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tmp_$name -o p -- python $R/bench.py "$@" --no-cpu-baseline --no-secondary > $O/${name}_bench.json 2>/dev/null
  f=$(find $O/tmp_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${name}_kernel_stats.csv
  rm -rf $O/tmp_$name
}
prof iv_sv_1attack --arch iv --steps 50 --warmup 5 --streams 1
prof gmm_1attack --steps 100 --warmup 10 --streams 1
tail -5 $O/pytest.log
