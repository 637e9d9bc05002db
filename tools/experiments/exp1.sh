#!/bin/bash
# diagnostic sweep: working-set aliasing, batch size, block size
B="timeout 120 python bench.py --cpu-seconds 0 --steps 10 --warmup 2"
short() { python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-40s %9.0f aligns/s  kernel %.3f ms  algo %.0f GB/s' % (sys.argv[1], d['value'], d['roofline']['kernel_ms'], d['roofline']['achieved']))" "$1"; }
for a in 0 256 32 4 1; do $B --debug-alias $a 2>/dev/null | short "alias=$a block=512"; done
for a in 0 32 1; do $B --block 256 --debug-alias $a 2>/dev/null | short "alias=$a block=256"; done
for b in 256 512 2048 4096; do $B --batch $b 2>/dev/null | short "batch=$b block=512"; done
$B --no-final-outputs 2>/dev/null | short "no final outputs"
