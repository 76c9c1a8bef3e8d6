#!/bin/bash
# i-vector iteration: tests of the i-vector path, kernel-trace stats of the single-attack SV chain (and B = 201), bench lines
R=$GRAFT_REPO_ROOT; tag=${1:-r05_iv}; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ivector.py tests/test_gpu_fullsize_ivector.py tests/test_gpu_baseline_configs.py -x -q -m gpu -s > $O/pytest_iv.log 2>&1; echo "pytest rc $?" >> $O/pytest_iv.log
cd /tmp && export TMPDIR=/tmp
prof() {
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tmp_$name -o p -- python $R/bench.py "$@" --no-cpu-baseline --no-secondary > $O/${name}_bench.json 2>/dev/null
  f=$(find $O/tmp_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${name}_kernel_stats.csv
  rm -rf $O/tmp_$name
}
prof iv_sv_1attack --arch iv --steps 50 --warmup 5 --streams 1
prof iv_osi_b201_1attack --arch iv --task OSI --speakers 10 --spd 200 --steps 20 --warmup 3 --streams 1
cd $R
timeout 600 python bench.py --arch iv --no-cpu-baseline > $O/iv_bench.json 2>/dev/null
tail -4 $O/pytest_iv.log
python - $O <<'PY'
import csv, json, sys
O = sys.argv[1]
for n in ("iv_sv_1attack", "iv_osi_b201_1attack"):
    try:
        tot = 0.0
        for r in csv.DictReader(open("%s/%s_kernel_stats.csv" % (O, n))):
            if int(r["Calls"]) > 10:
                print("%-28s %8.1f us" % (r["Name"].split("(")[0][-28:], float(r["AverageNs"]) / 1e3)); tot += float(r["AverageNs"]) / 1e3
        d = json.load(open("%s/%s_bench.json" % (O, n)))
        print(n, "sum %.1f us; ms/step %.4f value %.0f" % (tot, d["ms_per_step"], d["value"]))
    except Exception as ex: print(n, ex)
d = json.load(open(O + "/iv_bench.json")); print("iv 3 attacks", d["value"], d["single_attack"])
PY
