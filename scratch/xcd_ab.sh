#!/bin/bash
# A/B of the XCD-aware (strip, chunk) mapping of k_gmm_bx3
for i in 1 2; do
echo "xcd map on:";  python scratch/gmm_only.py
echo "xcd map off:"; FB_GMM_NO_XCD_MAP=1 python scratch/gmm_only.py
done
for t in 512 1024 2048; do echo "target $t on"; FB_GMM_TARGET_BLOCKS=$t python scratch/gmm_only.py; echo "target $t off"; FB_GMM_NO_XCD_MAP=1 FB_GMM_TARGET_BLOCKS=$t python scratch/gmm_only.py; done
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 200 --warmup 20 2>&1 | tail -1
FB_GMM_NO_XCD_MAP=1 python bench.py --steps 200 --warmup 20 2>&1 | tail -1
