#!/bin/bash
# k_gmm_fx2 (f16 two-term split, 3 MFMAs / 16 K) against k_gmm_bx3 (bf16 three-term split, 6 MFMAs) and k_gmm (f32)
for m in fx2 bx3 f32; do echo "mode $m:"; FB_GMM_MODE=$m python scratch/gmm_only.py; done
for t in 256 512 1024; do echo "fx2 target $t"; FB_GMM_TARGET_BLOCKS=$t python scratch/gmm_only.py; done
python scratch/bx_err.py 2>&1 | tail -8
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --streams 1 2>&1 | tail -1 | cut -c1-330
python bench.py --arch iv --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
