#!/bin/bash
# round 4, experiment 2: compile-time sweep direction; r03 library as the same-box baseline; what an L2 miss fetches
run() {  # name lib env... -- bench args
  name=$1; lib=$2; shift; shift
  envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs DVO_LIB_VARIANT=$lib python bench.py --no-extra-legs --cpu-seconds 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1])
print('%-10s %9d aligns/s  kernel %.3f ms  frac %.4f' % ('$name', round(d['value']), d['roofline']['kernel_ms'], d['roofline']['frac']))"
}
echo "== line fetch microbenchmark"
tools/experiments/r04_line_fetch.sh
echo "== quick parity"
python -m pytest tests/test_gpu_parity.py tests/test_gpu_packed_kernel.py tests/test_gpu_compact_now.py -m gpu -x -q 2>&1 | tail -3
for rep in 1 2; do
  echo "== C2 8192 pairs, rep $rep"
  run zigzag "" -- --batch 8192 --steps 30
  run nozz _nozz -- --batch 8192 --steps 30
  run r03 _r03 -- --batch 8192 --steps 30
  echo "== C2 1024 pairs, rep $rep"
  run zigzag "" -- --batch 1024 --steps 100
  run nozz _nozz -- --batch 1024 --steps 100
  run r03 _r03 -- --batch 1024 --steps 100
  echo "== C3 1024 pairs, rep $rep"
  run zigzag "" -- --width 1920 --height 1080 --levels 5 --batch 1024 --distinct 8 --steps 5 --warmup 1
  run r03 _r03 -- --width 1920 --height 1080 --levels 5 --batch 1024 --distinct 8 --steps 5 --warmup 1
done
