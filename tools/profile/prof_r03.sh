#!/bin/bash
# round-3 profile set: rocprofv3 kernel-trace stats of the bench commands (GMM headline with 1 / 3 attacks in flight,
# the reference-pipeline mode, i-vector SV spd=50 and OSI spd=200) and the default bench lines.
# usage: gpurun -- 'bash tools/profile/prof_r03.sh r03_a'   ->  gpurun_out/<tag>/
R=$GRAFT_REPO_ROOT; tag=$1; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
prof() {  # name, bench args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/tmp_$name -o p -- python $R/bench.py "$@" --no-cpu-baseline > $O/${name}_bench.json 2>/dev/null
  f=$(find $O/tmp_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${name}_kernel_stats.csv
  rm -rf $O/tmp_$name
}
prof gmm_1attack --steps 100 --warmup 10 --streams 1
prof gmm_3attacks --steps 100 --warmup 10
prof gmm_faithful_1attack --steps 100 --warmup 10 --streams 1 --faithful
prof iv_sv_1attack --arch iv --steps 50 --warmup 5 --streams 1
prof iv_osi_b201_1attack --arch iv --task OSI --speakers 10 --spd 200 --steps 20 --warmup 3 --streams 1
cd $R
python bench.py > $O/bench.json 2>/dev/null
python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2>/dev/null
python bench.py --faithful > $O/bench_faithful.json 2>/dev/null
python bench.py --arch iv > $O/iv_bench.json 2>/dev/null
python bench.py --arch iv --task OSI --speakers 10 --spd 200 --steps 30 --warmup 5 > $O/iv_osi_b201_bench.json 2>/dev/null
ls $O
