#!/bin/bash
# k_mfcc_f32: parity tests of the float32 front-end (both sample-load paths), kernel time in the single-attack chain, bench
R=$GRAFT_REPO_ROOT; tag=${1:-r05_mfcc}; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_mfcc_f32.py -x -q -m gpu > $O/pytest_mfcc.log 2>&1; echo "pytest rc $?" >> $O/pytest_mfcc.log
tail -6 $O/pytest_mfcc.log
cd /tmp && export TMPDIR=/tmp
prof() {
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tmp_$name -o p -- python $R/bench.py "$@" --no-cpu-baseline --no-secondary > $O/${name}_bench.json 2>/dev/null
  f=$(find $O/tmp_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${name}_kernel_stats.csv
  rm -rf $O/tmp_$name
}
prof gmm_1attack --steps 100 --warmup 10 --streams 1
FB_MFCC_RECORDS=1 prof gmm_1attack_records --steps 100 --warmup 10 --streams 1
cd $R
python - $O <<'PY'
import csv, json, sys
O = sys.argv[1]
for n in ("gmm_1attack", "gmm_1attack_records"):
    try:
        tot = 0.0
        for r in csv.DictReader(open("%s/%s_kernel_stats.csv" % (O, n))):
            if int(r["Calls"]) > 10:
                print("%-34s %8.1f us" % (r["Name"].split("(")[0][-34:], float(r["AverageNs"]) / 1e3)); tot += float(r["AverageNs"]) / 1e3
        d = json.load(open("%s/%s_bench.json" % (O, n)))
        print(n, "sum %.1f us; ms/step %.4f value %.0f" % (tot, d["ms_per_step"], d["value"]))
    except Exception as ex: print(n, ex)
PY
timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 200 --warmup 20 > $O/bench3.json 2>/dev/null
python -c "
import json; d=json.load(open('$O/bench3.json')); print('3 attacks', d['value'], d['secondary'].get('single_attack') if 'secondary' in d else '')"
bash tools/profile/mfcc_instrumented.sh gpurun_out/$tag/stamps > /dev/null 2>&1; cat gpurun_out/$tag/stamps/mfcc_stamps.txt | head -12
