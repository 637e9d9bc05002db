#!/bin/bash
# round 5: 4-byte reference points from which list length on?  DVO_POINTS4_FACTOR = 3 (product: lists >= 3x the LDS capacity), 1, 0 (always)
run() { python bench.py --no-extra-legs --cpu-seconds 0 "$@" 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%9.1f aligns/s  kernel %.3f ms  frac %.4f' % (d['value'], d['roofline'].get('kernel_ms') or 0, d['roofline']['frac']))"; }
for rep in 1 2; do for f in 3 1 0; do
  echo -n "factor $f c2 b8192 : "; DVO_POINTS4_FACTOR=$f run --batch 8192 --steps 30
  echo -n "factor $f c2 b1024 : "; DVO_POINTS4_FACTOR=$f run --batch 1024 --steps 100
  echo -n "factor $f c3 b1024 : "; DVO_POINTS4_FACTOR=$f run --width 1920 --height 1080 --levels 5 --batch 1024 --distinct 8 --steps 5 --warmup 1
done; done
