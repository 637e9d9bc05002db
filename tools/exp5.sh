#!/bin/bash
B="timeout 100 python bench.py --cpu-seconds 0 --steps 10 --warmup 2 --skip-hessian --block 512 --lds-point-bytes 155000"
short() { python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-52s %9.0f aligns/s  kernel %.3f ms' % (sys.argv[1], d['value'], d['roofline']['kernel_ms']))" "$1"; }
$B 2>&1 | tail -3 | cut -c1-300
for v in "" _t1x1 _t2x1 _t1x2 _t2x2 _t3x0 _t0x2; do
  DVO_LIB_VARIANT=$v $B 2>/dev/null | short "tile(log2 y x log2 x)=$v"
done
DVO_LIB_VARIANT=_t1x1 timeout 100 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
DVO_LIB_VARIANT=_t2x2 timeout 100 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
