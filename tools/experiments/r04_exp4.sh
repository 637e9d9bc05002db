#!/bin/bash
# vector diet x 4-byte points on every level (DVO_POINTS4_FACTOR): does relieving BOTH ceilings a little pay?  same-box A/B
# variants: base = product library, diet = tools/experiments/r04_patches/valu_diet_masks_full_rounds.patch built with make EXP=diet
run() { v=$1; f=$2; shift 2; lib=""; [ "$v" != "base" ] && lib="_$v"
  DVO_POINTS4_FACTOR=$f DVO_LIB_VARIANT=$lib python bench.py --no-extra-legs --cpu-seconds 0 "$@" 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('%9d aligns/s  kernel %.3f ms  frac %.4f' % (round(d['value']), r['kernel_ms'], r['frac']))"; }
for rep in 1 2; do for v in base diet; do for f in 3 1 0; do
  echo -n "$v pt4_factor=$f c2 b8192 : "; run $v $f --batch 8192 --steps 30
  echo -n "$v pt4_factor=$f c3 b1024 : "; run $v $f --width 1920 --height 1080 --levels 5 --batch 1024 --distinct 8 --steps 5 --warmup 1
done; done; done
