#!/bin/bash
# Instrumented build of k_gmm_finalize_loss_update (gmm_kernels.hip, -DFB_FIN_STAMP): wall_clock64 stamps of every
# workgroup -- where the launch goes.  usage (GPU box): tools/profile/fin_instrumented.sh [outdir]
set -e
cd "$(dirname "$0")/../.."
OUT=${1:-gpurun_out/fin}
mkdir -p "$OUT"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-unused-function"
OBJS=$(ls fakebob_amd/build/*.o | grep -v gmm_kernels)
/opt/rocm/bin/hipcc $FLAGS -DFB_FIN_STAMP -c fakebob_amd/csrc/gmm_kernels.hip -o "$OUT/fin_stamp.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/lib_stamp.so" "$OUT/fin_stamp.o" $OBJS
FAKEBOB_HIP_LIB="$PWD/$OUT/lib_stamp.so" python tools/profile/fin_instrumented.py | tee "$OUT/fin_stamps.txt"
