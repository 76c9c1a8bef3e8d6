#!/bin/bash
# Instrumented build of k_vad_delta_cmvn_p (frontend_kernels.hip, -DFB_VAD_STAMP): wall_clock64 stamps of every
# workgroup -- where the launch goes.  usage (GPU box): tools/profile/vad_instrumented.sh [outdir]
set -e
cd "$(dirname "$0")/../.."
OUT=${1:-gpurun_out/vad}
mkdir -p "$OUT"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-unused-function"
OBJS=$(ls fakebob_amd/build/*.o | grep -v frontend_kernels)
/opt/rocm/bin/hipcc $FLAGS -DFB_VAD_STAMP -c fakebob_amd/csrc/frontend_kernels.hip -o "$OUT/vad_stamp.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/lib_stamp.so" "$OUT/vad_stamp.o" $OBJS
FAKEBOB_HIP_LIB="$PWD/$OUT/lib_stamp.so" python tools/profile/vad_instrumented.py | tee "$OUT/vad_stamps.txt"
