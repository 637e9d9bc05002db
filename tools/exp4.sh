#!/bin/bash
B="timeout 100 python bench.py --cpu-seconds 0 --steps 10 --warmup 2 --skip-hessian"
short() { python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-52s %9.0f aligns/s  kernel %.3f ms' % (sys.argv[1], d['value'], d['roofline']['kernel_ms']))" "$1"; }
for blk in 256 512 1024; do
  $B --block $blk --lds-point-bytes -1 2>/dev/null | short "block=$blk lds=none"
  $B --block $blk 2>/dev/null | short "block=$blk lds=auto"
done
$B --block 512 --lds-point-bytes 155000 2>/dev/null | short "block=512 lds=155000 (1 WG/CU)"
$B --block 256 --lds-point-bytes 77000 2>/dev/null | short "block=256 lds=77000 (2 WG/CU)"
$B --block 512 --batch 512 2>/dev/null | short "block=512 lds=auto batch=512"
$B --block 1024 --batch 256 2>/dev/null | short "block=1024 lds=auto batch=256"
