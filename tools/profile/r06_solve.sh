#!/bin/bash
# round 6: the solve kernels after a change -- i-vector tests (both kernels), the fenced build, the serial chain's stamps, the
# SV / B = 201 lines with one attack in flight and the SV line with three
R=$GRAFT_REPO_ROOT; tag=${1:-r06_solve}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_ivector.py tests/test_gpu_fullsize_ivector.py tests/test_gpu_fenced.py -x -q 2>&1 | tail -3
FB_IV_SOLVE=ll timeout 900 python -m pytest tests/test_gpu_ivector.py tests/test_gpu_fullsize_ivector.py -x -q 2>&1 | tail -2
bash tools/profile/rw_instrumented.sh gpurun_out/$tag/rw 2>&1 | tail -16
rm -f gpurun_out/$tag/rw/*.o gpurun_out/$tag/rw/*.so
for v in tree; do
  timeout 300 python bench.py --arch iv --steps 40 --warmup 5 --streams 1 --no-cpu-baseline > $O/sv_$v.json 2>$O/sv_$v.err
  timeout 300 python bench.py --arch iv --task OSI --speakers 10 --spd 200 --steps 10 --warmup 3 --streams 1 --no-cpu-baseline > $O/b201_$v.json 2>$O/b201_$v.err
  timeout 300 python bench.py --arch iv --steps 40 --warmup 5 --streams 3 --no-cpu-baseline > $O/sv3_$v.json 2>$O/sv3_$v.err
  python - $O $v <<'PY'
import json,sys
for n in ("sv","b201","sv3"):
    try:
        d=json.load(open("%s/%s_%s.json"%(sys.argv[1],n,sys.argv[2])))
        print("%s %s: %.0f it/s (%.3f ms), solve launch %.1f us, contraction %.1f us" % (sys.argv[2], n, d["value"], d["ms_per_step"], 1e3*d["roofline_solve"]["avg_launch_ms"], 1e3*d["roofline_contraction"]["avg_launch_ms"]))
    except Exception as ex: print(n, "FAILED", ex)
PY
done
