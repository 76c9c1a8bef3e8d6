#!/bin/bash
# Instrumented build of k_iv_solve_rw (ivector_solve.hip, -DFB_RW_STAMP): wall_clock64 stamps of matrix 0's block rows --
# where the serial chain of the row-wise factorisation spends its time.  usage (GPU box): tools/profile/rw_instrumented.sh [outdir]
set -e
cd "$(dirname "$0")/../.."
OUT=${1:-gpurun_out/rw}
mkdir -p "$OUT"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-unused-function"
OBJS=$(ls fakebob_amd/build/*.o | grep -v ivector_solve)
/opt/rocm/bin/hipcc $FLAGS -DFB_RW_STAMP -c fakebob_amd/csrc/ivector_solve.hip -o "$OUT/solve_stamp.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/lib_stamp.so" "$OUT/solve_stamp.o" $OBJS
FAKEBOB_HIP_LIB="$PWD/$OUT/lib_stamp.so" FB_IV_SOLVE=rw python tools/profile/rw_instrumented.py | tee "$OUT/rw_stamps.txt"
