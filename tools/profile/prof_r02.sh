#!/bin/bash
# round-2 profile set: kernel-trace stats (1 attack / default 3 in flight, GMM and i-vector), HBM-traffic PMC passes
# (separate runs, --kernel-trace only), default bench lines.  usage: gpurun -- 'bash tools/profile/prof_r02.sh r02_a'
R=$GRAFT_REPO_ROOT; tag=$1; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/s1 -o p -- python $R/bench.py --steps 100 --warmup 10 --streams 1 --no-cpu-baseline > $O/bench_1attack.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/s3 -o p -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_prof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/iv1 -o p -- python $R/bench.py --arch iv --steps 50 --warmup 5 --streams 1 --no-cpu-baseline > $O/iv_bench_1attack.json 2>/dev/null
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc_$name -o p -- python $R/bench.py --steps 20 --warmup 3 --streams 1 --no-cpu-baseline > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmciv_$name -o p -- python $R/bench.py --arch iv --steps 10 --warmup 2 --streams 1 --no-cpu-baseline > /dev/null 2>&1
done
cd $R
python bench.py > $O/bench.json 2>/dev/null
python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2>/dev/null
python bench.py --arch iv > $O/iv_bench.json 2>/dev/null
python - $O <<'PY'
import csv, collections, glob, json, sys
O = sys.argv[1]
out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum (three separate passes each) -- "
                 "python bench.py [--arch iv] --streams 1 --no-cpu-baseline; per-launch averages; FETCH_SIZE/WRITE_SIZE in KiB as reported",
       "correction": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads -> "
                     "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024", "kernels": {}}
for pre in ("pmc_", "pmciv_"):
    res = collections.defaultdict(dict)
    for name in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum"):
        f = glob.glob("%s/%s%s/**/*counter_collection.csv" % (O, pre, name), recursive=True)
        if not f: continue
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f[0])):
            k = r["Kernel_Name"].split("(")[0]
            if k.startswith("void "): k = k[5:]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
        for k in agg:
            for c, v in agg[k].items(): res[k][c] = v / cnt[(k, c)]
    for k, v in res.items():
        e = {"FETCH_SIZE_KiB": round(v.get("FETCH_SIZE", 0.0), 1), "WRITE_SIZE_KiB": round(v.get("WRITE_SIZE", 0.0), 1)}
        e["hbm_bytes_per_launch"] = int((2 * e["FETCH_SIZE_KiB"] + e["WRITE_SIZE_KiB"]) * 1024)
        h, m = v.get("TCC_HIT_sum", 0.0), v.get("TCC_MISS_sum", 0.0)
        e["l2_hit_rate"] = round(h / (h + m), 3) if h + m > 0 else None
        out["kernels"][k] = e
ks = out["kernels"]
lin = [k for k in ks if k.startswith("k_iv_contract_dma<true>")]; quad = [k for k in ks if k.startswith("k_iv_contract_dma<false>")]
if lin and quad:
    out["kernels"]["k_iv_contract_dma<lin>+<quad>"] = {"hbm_bytes_per_launch": ks[lin[0]]["hbm_bytes_per_launch"] + ks[quad[0]]["hbm_bytes_per_launch"]}
json.dump(out, open(O + "/traffic.json", "w"), indent=1)
for k, v in sorted(ks.items()): print(k, v)
PY
for d in s1 s3 iv1; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_$d.csv; done
rm -rf $O/s1 $O/s3 $O/iv1 $O/pmc_* $O/pmciv_*
ls $O
