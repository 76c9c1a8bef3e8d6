#!/bin/bash
# phases of k_iv_solve_packed (timing probes: the results are wrong but finite, nothing stops the attack)
for f in "" "-DFB_SOLVE_PROBE_SETUP" "-DFB_SOLVE_PROBE_CHOL"; do
  FB_EXTRA_HIPCC_FLAGS="$f" python -c "from fakebob_amd import build; build.build(force=True)" 2>&1 | grep -i " error"
  echo "== $f"; bash scratch/iv_prof.sh x 2>&1 | grep -E "solve|it/s"
done
