#!/bin/bash
# round 4, experiment 3: speed proxy of a one-byte-per-pixel now form (5 x 16 interior pixels per 128-byte line; data wrong, access pattern right)
run() {  # name lib env... -- bench args
  name=$1; lib=$2; shift; shift
  envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs DVO_LIB_VARIANT=$lib python bench.py --no-extra-legs --cpu-seconds 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1])
print('%-10s %9d aligns/s  kernel %.3f ms  frac %.4f' % ('$name', round(d['value']), d['roofline']['kernel_ms'], d['roofline']['frac']))"
}
for rep in 1 2; do
  echo "== C2 8192 pairs, rep $rep"
  run nozz _nozz -- --batch 8192 --steps 30
  run b1 _b1 -- --batch 8192 --steps 30
  run b1_pt4 _b1 DVO_POINTS4_FACTOR=1 -- --batch 8192 --steps 30
  run b1w3 _b1w3 DVO_WGS_PER_CU=3 -- --batch 8192 --steps 30
  run b1w3_pt4 _b1w3 DVO_WGS_PER_CU=3 DVO_POINTS4_FACTOR=1 -- --batch 8192 --steps 30
  echo "== C3 1024 pairs, rep $rep"
  run nozz _nozz -- --width 1920 --height 1080 --levels 5 --batch 1024 --distinct 8 --steps 5 --warmup 1
  run b1 _b1 -- --width 1920 --height 1080 --levels 5 --batch 1024 --distinct 8 --steps 5 --warmup 1
done
echo "== PMC: requests and L2 hits, 4096 pairs"
for v in "nozz:_nozz" "b1:_b1"; do
  n=${v%%:*}; lib=${v#*:}
  DVO_LIB_VARIANT=$lib tools/pmc_one.sh r04e3_$n "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum" --batch 4096 2>&1 | sed "s/^/$n /"
done
DVO_LIB_VARIANT=_b1 DVO_POINTS4_FACTOR=1 tools/pmc_one.sh r04e3_b1pt4 "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum" --batch 4096 2>&1 | sed "s/^/b1pt4 /"
