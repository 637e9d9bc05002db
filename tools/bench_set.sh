#!/bin/bash
# usage: tools/bench_set.sh <tag> [extra bench args] -- the standard set of bench points for A/B runs: b1024, alias1, b256, b32
TAG=$1; shift
mkdir -p gpurun_out
for cfg in "b1024:--batch 1024" "alias1:--batch 1024 --debug-alias 1" "b256:--batch 256" "b32:--batch 32"; do
  name=${cfg%%:*}; args=${cfg#*:}
  python bench.py --no-frames-leg --cpu-seconds 0 $args "$@" > gpurun_out/${TAG}_$name.json 2>>gpurun_out/${TAG}_err.log
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/${TAG}_$name.json"))
    print("${TAG}_$name", round(d["value"]), "aligns/s  step %.3f ms  kernel %.3f ms  frac %.4f" % (d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"]), d.get("parity_check"))
except Exception as e:
    print("${TAG}_$name FAILED", e)
PY
done
