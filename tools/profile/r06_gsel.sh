#!/bin/bash
# round 6: the threshold gselect -- its tests, the i-vector suites, the i-vector bench line with and without it, kernel stats
R=$GRAFT_REPO_ROOT; tag=${1:-r06_gsel}; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_gselect.py -x -q -s > $O/pytest_gsel.log 2>&1; echo "rc $?" >> $O/pytest_gsel.log
grep -v "^\.\+$" $O/pytest_gsel.log | tail -25
timeout 900 python -m pytest tests/test_gpu_ivector.py tests/test_gpu_fullsize_ivector.py -x -q > $O/pytest_iv.log 2>&1; echo "rc $?" >> $O/pytest_iv.log
tail -5 $O/pytest_iv.log
for m in new dump; do
  unset FB_IV_GSEL_DUMP FB_GSEL_NARROW; [ $m = dump ] && export FB_IV_GSEL_DUMP=1
  timeout 300 python bench.py --arch iv --steps 30 --warmup 5 --no-cpu-baseline > $O/iv_$m.json 2>$O/iv_$m.err
  python - $O/iv_$m.json $m <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[2], "value %.0f single %.0f (%.3f ms) solve %.1f us" % (d["value"], d["single_attack"]["value"], d["single_attack"]["ms_per_step"], 1e3*d["roofline_solve"]["avg_launch_ms"]))
except Exception as ex: print(sys.argv[2], "FAILED", ex)
PY
done
unset FB_IV_GSEL_DUMP FB_GSEL_NARROW
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tmp_iv -o p -- python $R/bench.py --arch iv --steps 50 --warmup 5 --streams 1 --no-cpu-baseline --no-secondary > $O/iv_sv_1attack_bench.json 2>/dev/null
f=$(find $O/tmp_iv -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/iv_sv_1attack_kernel_stats.csv
rm -rf $O/tmp_iv
python - $O/iv_sv_1attack_kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:26]:
    if int(r['Calls']) > 10: print('   %-44s %7.1f us x %s' % (r['Name'].split('(')[0][-44:], float(r['AverageNs']) / 1e3, r['Calls']))
PY
