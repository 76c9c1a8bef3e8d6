#!/bin/bash
# k_mfcc_r4 vs k_mfcc_r16 (FB_MFCC=r4|r16) in the default bench
for i in 1 2; do
for m in r4 r16; do echo -n "FB_MFCC=$m: "; FB_MFCC=$m python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1))"; done
done
for m in r4 r16; do echo -n "FB_MFCC=$m 1 stream: "; FB_MFCC=$m python bench.py --steps 200 --warmup 20 --no-cpu-baseline --streams 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1))"; done
for m in r4 r16; do echo -n "FB_MFCC=$m iv: "; FB_MFCC=$m python bench.py --arch iv --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1))"; done
