#!/bin/bash
# round-6 profile set on the GPU box: rocprofv3 kernel-trace stats of the bench commands (GMM headline with 1 / 3
# attacks in flight, three products / the F6 class forced, realistic enrolment, the reference-pipeline mode, GMM CSI, i-vector SV
# spd=50 and OSI spd=200), the HBM-traffic PMC passes (separate runs, --kernel-trace only, one counter group per pass)
# and the default bench lines.  The GMM PMC passes run the chain of a shared GPU (--chain unfused: the headline's launch shapes).  traffic.json records the hash of the kernel sources it was taken on: bench.py reports
# roofline.traffic only while that hash is the build's (bench.kernel_source_hash()).
# usage: gpurun -- 'bash tools/profile/prof_r06.sh r06_a [commit]'   ->  gpurun_out/<tag>/
R=$GRAFT_REPO_ROOT; tag=$1; commit=${2:-unknown}; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
prof() {  # name, bench args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/tmp_$name -o p -- python $R/bench.py "$@" --no-cpu-baseline --no-secondary > $O/${name}_bench.json 2>/dev/null
  f=$(find $O/tmp_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${name}_kernel_stats.csv
  rm -rf $O/tmp_$name
}
prof gmm_1attack --steps 100 --warmup 10 --streams 1
prof gmm_3attacks --steps 100 --warmup 10
prof gmm_1attack_shared_gpu_chain --steps 100 --warmup 10 --streams 1 --chain unfused   # the headline's launch shapes, alone on the chip
FB_GMM_DELTA_P=3 prof gmm_p3_1attack --steps 100 --warmup 10 --streams 1
FB_GMM_DELTA_P=6 prof gmm_f6_1attack --steps 100 --warmup 10 --streams 1
prof gmm_realistic_1attack --steps 100 --warmup 10 --streams 1 --enrol realistic
prof gmm_faithful_1attack --steps 100 --warmup 10 --streams 1 --faithful
prof gmm_csi_1attack --steps 100 --warmup 10 --streams 1 --task CSI
prof iv_sv_1attack --arch iv --steps 50 --warmup 5 --streams 1
prof iv_osi_b201_1attack --arch iv --task OSI --speakers 10 --spd 200 --steps 20 --warmup 3 --streams 1
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc_$name -o p -- python $R/bench.py --steps 20 --warmup 3 --streams 1 --chain unfused --no-cpu-baseline --no-secondary > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmciv_$name -o p -- python $R/bench.py --arch iv --steps 10 --warmup 2 --streams 1 --no-cpu-baseline > /dev/null 2>&1
done
cd $R
python - $O $commit <<'PY'
import csv, collections, glob, json, sys, importlib.util
O, commit = sys.argv[1], sys.argv[2]
spec = importlib.util.spec_from_file_location("fb_bench", "bench.py"); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum (three separate passes each) -- "
                 "python bench.py [--arch iv] --streams 1 --no-cpu-baseline --no-secondary; per-launch averages; FETCH_SIZE/WRITE_SIZE in KiB as reported",
       "correction": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads -> "
                     "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024",
       "kernel_source_sha16": bench.kernel_source_hash(), "commit": commit, "kernels": {}}
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
both = [k for k in ks if k.startswith("k_iv_contract_both")]
if both:  # (one launch since the end of round 4; the key keeps its name: bench.py reads it)
    out["kernels"]["k_iv_contract_dma<lin>+<quad>"] = {"hbm_bytes_per_launch": ks[both[0]]["hbm_bytes_per_launch"]}
elif lin and quad:
    out["kernels"]["k_iv_contract_dma<lin>+<quad>"] = {"hbm_bytes_per_launch": ks[lin[0]]["hbm_bytes_per_launch"] + ks[quad[0]]["hbm_bytes_per_launch"]}
json.dump(out, open(O + "/traffic.json", "w"), indent=1)
for k, v in sorted(ks.items()): print(k, v)
PY
rm -rf $O/pmc_* $O/pmciv_*
# the default lines read the traffic file just written (same sources): put it where bench.py looks
cp $O/traffic.json $R/profiles/r06_traffic.json
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2>/dev/null
python bench.py --faithful --no-secondary > $O/bench_faithful.json 2>/dev/null
python bench.py --task CSI --no-secondary > $O/bench_gmm_csi.json 2>/dev/null
python bench.py --enrol realistic --no-secondary > $O/bench_realistic.json 2>/dev/null
python bench.py --arch iv > $O/iv_bench.json 2>/dev/null
python bench.py --arch iv --task OSI --speakers 10 --spd 200 --steps 30 --warmup 5 > $O/iv_osi_b201_bench.json 2>/dev/null
ls $O
