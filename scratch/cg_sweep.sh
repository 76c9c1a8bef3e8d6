#!/bin/bash
for f in "-DFB_CG_KC=8" "-DFB_CG_KC=16" "-DFB_CG_KC=32"; do
  FB_EXTRA_HIPCC_FLAGS="$f" python -c "from fakebob_amd import build; build.build(force=True)" 2>&1 | grep -i " error"
  echo "== $f"; timeout 200 scratch/iv_prof.sh 2>&1 | grep -E "contract|it/s"
done
