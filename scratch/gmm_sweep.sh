#!/bin/bash
for tb in 512 768 1024 1280 1536 2048; do
  echo -n "target_blocks=$tb "; FB_GMM_TARGET_BLOCKS=$tb python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('it/s %.1f  gmm_ms %.4f  TF %.1f' % (d['value'], d['roofline']['avg_launch_ms'], d['roofline']['achieved']))"
done
