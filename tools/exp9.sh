#!/bin/bash
B="timeout 100 python bench.py --cpu-seconds 0 --steps 10 --warmup 2"
short() { python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-52s %9.0f aligns/s  kernel %.3f ms' % (sys.argv[1], d['value'], d['roofline']['kernel_ms']))" "$1"; }
for v in "" _inl; do
 for blk in 256 512 1024; do
  DVO_LIB_VARIANT=$v $B --block $blk 2>/dev/null | short "lib=$v block=$blk lds=auto"
 done
 DVO_LIB_VARIANT=$v $B --block 512 --lds-point-bytes 77000 2>/dev/null | short "lib=$v block=512 lds=77000 (2 WG/CU)"
 DVO_LIB_VARIANT=$v $B --block 256 --lds-point-bytes 38000 2>/dev/null | short "lib=$v block=256 lds=38000 (4 WG/CU)"
done
timeout 100 python tools/exp7.py 256 1024 155000 1 2>&1 | grep -v amdgpu
