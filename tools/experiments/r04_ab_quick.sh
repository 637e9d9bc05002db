#!/bin/bash
# same-box A/B of the product library against variants: tools/experiments/r04_ab_quick.sh "new base ..." [reps]
VARS=${1:-"new base"}; REPS=${2:-2}
run() { lib=""; [ "$1" != "new" ] && lib="_$1"; shift
  DVO_LIB_VARIANT=$lib python bench.py --no-extra-legs --cpu-seconds 0 "$@" 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%9d aligns/s  kernel %.3f ms  frac %.4f' % (round(d['value']), d['roofline']['kernel_ms'], d['roofline']['frac']))"; }
for rep in $(seq $REPS); do for v in $VARS; do
  echo -n "$v c2 b8192 : "; run $v --batch 8192 --steps 30
  echo -n "$v c2 b1024 : "; run $v --batch 1024 --steps 100
  echo -n "$v c3 b1024 : "; run $v --width 1920 --height 1080 --levels 5 --batch 1024 --distinct 8 --steps 5 --warmup 1
done; done
