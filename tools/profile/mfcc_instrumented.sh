#!/bin/bash
# Instrumented build of k_mfcc_f32 (frontend_f32_kernels.hip, -DFB_MFCC_STAMP): wall_clock64 stamps of the 16 waves of four
# workgroups -- where a wave spends the launch.  usage (GPU box): tools/profile/mfcc_instrumented.sh [outdir]
set -e
cd "$(dirname "$0")/../.."
OUT=${1:-gpurun_out/mfcc}
mkdir -p "$OUT"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-unused-function"
OBJS=$(ls fakebob_amd/build/*.o | grep -v frontend_f32_kernels)
/opt/rocm/bin/hipcc $FLAGS -DFB_MFCC_STAMP -c fakebob_amd/csrc/frontend_f32_kernels.hip -o "$OUT/mfcc_stamp.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/lib_stamp.so" "$OUT/mfcc_stamp.o" $OBJS
FAKEBOB_HIP_LIB="$PWD/$OUT/lib_stamp.so" python tools/profile/mfcc_instrumented.py | tee "$OUT/mfcc_stamps.txt"
