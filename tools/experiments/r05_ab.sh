#!/bin/bash
# round 5: same-box A/B of the product library against the round-4 tree (make EXP=r4base from the round-4 sources) on the
# latency-bound shapes and the default line, plus the stamps anatomy of both: tools/experiments/r05_ab.sh [reps] ["variants"] ["stamps variants"]
REPS=${1:-2}; VARS=${2:-"new r4base"}; STAMPV=${3:-"_stamps _stamps_r4base"}
run() { lib=""; [ "$1" != "new" ] && lib="_$1"; shift
  DVO_LIB_VARIANT=$lib python bench.py --no-extra-legs --cpu-seconds 0 "$@" 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%9.1f aligns/s  %.4f ms/step  kernel %.3f ms  frac %.4f' % (d['value'], d['ms_per_step'], d['roofline'].get('kernel_ms') or 0, d['roofline']['frac']))"; }
for rep in $(seq $REPS); do for v in $VARS; do
  echo -n "$v c2 b32    : "; run $v --batch 32 --steps 400 --warmup 20
  echo -n "$v c2 b64    : "; run $v --batch 64 --steps 400 --warmup 20
  echo -n "$v c2 b256   : "; run $v --batch 256 --steps 300 --warmup 20
  echo -n "$v c2 b1024  : "; run $v --batch 1024 --steps 100
  echo -n "$v c2 b8192  : "; run $v --batch 8192 --steps 30
  echo -n "$v c3 b1024  : "; run $v --width 1920 --height 1080 --levels 5 --batch 1024 --distinct 8 --steps 5 --warmup 1
  echo -n "$v tiled C5  : "; run $v --mode tiled
  echo -n "$v tiled C2  : "; run $v --mode tiled --width 640 --height 480 --levels 4
done; done
for v in $STAMPV; do
  for cfg in "2048 0" "32 0" "256 0"; do
    echo "== stamps $v B/block = $cfg"; DVO_STAMPS_VARIANT=$v PREP=1 python tools/experiments/exp_stamps2.py $cfg 2>&1 | tail -6
  done
done
