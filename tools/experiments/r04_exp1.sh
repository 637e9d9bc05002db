#!/bin/bash
# round 4, experiment 1: backward sweep on odd iterations (zigzag) vs forward only, three 256-thread workgroups per CU, 4-byte points at C2
# usage (GPU box): tools/experiments/r04_exp1.sh > gpurun_out/r04_exp1.txt
run() {  # name lib env... -- bench args
  name=$1; lib=$2; shift; shift
  envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs DVO_LIB_VARIANT=$lib python bench.py --no-extra-legs --cpu-seconds 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1])
print('%-10s %9d aligns/s  kernel %.3f ms  frac %.4f  parity %s' % ('$name', round(d['value']), d['roofline']['kernel_ms'], d['roofline']['frac'], d.get('parity_check',{}).get('pass')))"
}
echo "== quick parity"
python -m pytest tests/test_gpu_parity.py tests/test_gpu_packed_kernel.py tests/test_gpu_compact_now.py -m gpu -x -q 2>&1 | tail -3
for rep in 1 2; do
  echo "== C2 8192 pairs, rep $rep"
  run zigzag "" -- --batch 8192 --steps 30
  run nozz _nozz -- --batch 8192 --steps 30
  run zz_pt4 "" DVO_POINTS4_FACTOR=1 -- --batch 8192 --steps 30
  run zz_w3 _w3 DVO_WGS_PER_CU=3 -- --batch 8192 --steps 30
  run zz_w3_pt4 _w3 DVO_WGS_PER_CU=3 DVO_POINTS4_FACTOR=1 -- --batch 8192 --steps 30
  echo "== C2 1024 pairs, rep $rep"
  run zigzag "" -- --batch 1024 --steps 100
  run nozz _nozz -- --batch 1024 --steps 100
  echo "== C3 1024 pairs, rep $rep"
  run zigzag "" -- --width 1920 --height 1080 --levels 5 --batch 1024 --distinct 8 --steps 5 --warmup 1
  run nozz _nozz -- --width 1920 --height 1080 --levels 5 --batch 1024 --distinct 8 --steps 5 --warmup 1
done
echo "== PMC: requests and L2 hits, 4096 pairs"
for v in "zigzag:" "nozz:_nozz"; do
  n=${v%%:*}; lib=${v#*:}
  DVO_LIB_VARIANT=$lib tools/pmc_one.sh r04e1_$n "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum" --batch 4096 2>&1 | sed "s/^/$n /"
done
