#!/bin/bash
B="timeout 100 python bench.py --cpu-seconds 0 --steps 10 --warmup 2 --skip-hessian"
short() { python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-52s %9.0f aligns/s  kernel %.3f ms' % (sys.argv[1], d['value'], d['roofline']['kernel_ms']))" "$1"; }
$B --block 512 --lds-point-bytes 155000 2>/dev/null | short "block=512 lds=155000 (1 WG/CU)"
$B --block 512 2>/dev/null | short "block=512 lds=auto(76K)"
$B --block 256 --lds-point-bytes 77000 2>/dev/null | short "block=256 lds=77000"
$B --block 256 2>/dev/null | short "block=256 lds=auto(36K)"
$B --block 1024 2>/dev/null | short "block=1024 lds=auto(152K)"
$B --block 512 --lds-point-bytes 155000 --debug-alias 1 2>/dev/null | short "block=512 lds=155000 alias=1"
timeout 100 python tools/exp2.py 1 512 0 2>&1 | grep -v amdgpu | grep -E "N<=     1|N<=100000|full"
