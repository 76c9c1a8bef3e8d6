#!/bin/bash
# k_iv_contract_dma ring depth
for f in "-DFB_CD_S=3" "-DFB_CD_S=4" "-DFB_CD_S=5"; do
  FB_EXTRA_HIPCC_FLAGS="$f" python -c "from fakebob_amd import build; build.build(force=True)" 2>&1 | grep -i " error"
  echo "== $f"; bash scratch/iv_prof.sh x 2>&1 | grep -E "contract|it/s"
  python bench.py --arch iv --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-140
done
