#!/bin/bash
# A/B of non-temporal loads of the streamed reference points (ntp), non-temporal stores of the final outputs (ntf), both (ntpf)
# against the product library: C2 at 8192 pairs and C3 at 256 pairs, twice, interleaved.  make EXP=ntp EXPDEFS=-DDVO_NT_POINTS=1 ...
mkdir -p gpurun_out
for rep in 1 2; do
  for v in base ntp ntf ntpf; do
    lib=""; [ "$v" != "base" ] && lib="_$v"
    for cfg in "c2:--batch 8192 --steps 30" "c3:--width 1920 --height 1080 --levels 5 --batch 256 --distinct 8 --steps 5 --warmup 1"; do
      name=${cfg%%:*}; args=${cfg#*:}
      DVO_LIB_VARIANT=$lib python bench.py --no-extra-legs --no-frames-leg --cpu-seconds 0 $args > gpurun_out/nt_${v}_$name.json 2>>gpurun_out/nt_err.log
      python - <<PY
import json
try:
    d=json.load(open("gpurun_out/nt_${v}_$name.json"))
    print("%-6s %-3s rep$rep %8d aligns/s  kernel %.3f ms  frac %.4f" % ("$v", "$name", round(d["value"]), d["roofline"]["kernel_ms"], d["roofline"]["frac"]))
except Exception as e:
    print("$v $name FAILED", e)
PY
    done
  done
done
