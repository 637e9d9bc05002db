#!/bin/bash
# sweep: register budget (library variant) x block x points-in-flight x hessian
B="timeout 100 python bench.py --cpu-seconds 0 --steps 10 --warmup 2"
short() { python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-52s %9.0f aligns/s  kernel %.3f ms' % (sys.argv[1], d['value'], d['roofline']['kernel_ms']))" "$1"; }
for v in "" _w2 _w3 _w4 _w5; do
 for blk in 128 256 512; do
  for u in 1 4; do
    DVO_LIB_VARIANT=$v $B --block $blk --inflight $u 2>/dev/null | short "lib=$v block=$blk U=$u H=1"
  done
  DVO_LIB_VARIANT=$v $B --block $blk --inflight 4 --skip-hessian 2>/dev/null | short "lib=$v block=$blk U=4 H=0"
 done
done
