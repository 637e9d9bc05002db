#!/bin/bash
# round 5: coarse levels' ranks from an LDS copy of the level (default) vs gathered from HBM / L2 (DVO_RANKS_LDS=off), same box, interleaved
run() { python bench.py --no-extra-legs --cpu-seconds 0 "$@" 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%9.1f aligns/s  kernel %.3f ms  frac %.4f' % (d['value'], d['roofline'].get('kernel_ms') or 0, d['roofline']['frac']))"; }
for rep in 1 2; do for mode in on off; do
  echo -n "$mode c2 b1024 : "; DVO_RANKS_LDS=$mode run --batch 1024 --steps 100
  echo -n "$mode c2 b8192 : "; DVO_RANKS_LDS=$mode run --batch 8192 --steps 30
  echo -n "$mode c2 b256  : "; DVO_RANKS_LDS=$mode run --batch 256 --steps 200
  echo -n "$mode c3 b1024 : "; DVO_RANKS_LDS=$mode run --width 1920 --height 1080 --levels 5 --batch 1024 --distinct 8 --steps 5 --warmup 1
  echo -n "$mode refdefault 320x240x4x50 b1024 : "; DVO_RANKS_LDS=$mode run --width 320 --height 240 --iters 50 --batch 1024 --steps 20
done; done
