#!/bin/bash
for v in ""; do
  for a in "--width 4096 --height 3072 --levels 5" "--width 1920 --height 1080 --levels 5" "--width 640 --height 480 --levels 4"; do
  DVO_LIB_VARIANT=$v python bench.py --mode tiled --cpu-seconds 1 $a 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('tiled [$v] $a: %.1f aligns/s  %.3f ms/align  %.2f us/iter  level-0 step %.2f us  graph=%s parity=%s' % (d['value'], d['ms_per_step'], d['config']['us_per_iteration'], 1e3*d['roofline']['kernel_ms'], d['config'].get('graph_replayed'), d.get('parity_check',{}).get('pass')))"
  done
done
