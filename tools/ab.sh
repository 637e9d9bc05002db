#!/bin/bash
# usage: tools/ab.sh "<variant> <variant> ..." [bench args]   -- same-box A/B of library builds (make EXP=...): "" or "base" = the
# product library; every variant runs the two bench points b1024 and alias1 (no L2 misses), twice, interleaved
VARS=$1; shift
mkdir -p gpurun_out
for rep in 1 2; do
  for v in $VARS; do
    lib=""; [ "$v" != "base" ] && lib="_$v"
    for cfg in "b1024:--batch 1024" "alias1:--batch 1024 --debug-alias 1"; do
      name=${cfg%%:*}; args=${cfg#*:}
      DVO_LIB_VARIANT=$lib python bench.py --no-frames-leg --cpu-seconds 0 $args "$@" > gpurun_out/ab_${v}_$name.json 2>>gpurun_out/ab_err.log
      python - <<PY
import json
try:
    d=json.load(open("gpurun_out/ab_${v}_$name.json"))
    print("%-14s %-7s rep$rep %8d aligns/s  kernel %.3f ms  frac %.4f" % ("$v", "$name", round(d["value"]), d["roofline"]["kernel_ms"], d["roofline"]["frac"]))
except Exception as e:
    print("$v $name FAILED", e)
PY
    done
  done
done
