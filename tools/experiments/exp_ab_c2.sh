#!/bin/bash
# same-box A/B of library builds on the headline configuration (C2, 8192 pairs) and C3 (256 pairs): tools/experiments/exp_ab_c2.sh "base new ..."
# ("new" = the product library; others = make EXP=<name> builds), three interleaved repetitions
VARS=${1:-"base new"}
for rep in 1 2 3; do
  for v in $VARS; do
    lib=""; [ "$v" != "new" ] && lib="_$v"
    for cfg in "c2:--batch 8192 --steps 30" "c3:--width 1920 --height 1080 --levels 5 --batch 256 --distinct 8 --steps 5 --warmup 1"; do
      name=${cfg%%:*}; args=${cfg#*:}
      DVO_LIB_VARIANT=$lib python bench.py --no-extra-legs --cpu-seconds 0 $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-6s %-3s rep$rep %8d aligns/s  kernel %.3f ms  frac %.4f' % ('$v', '$name', round(d['value']), d['roofline']['kernel_ms'], d['roofline']['frac']))"
    done
  done
done
