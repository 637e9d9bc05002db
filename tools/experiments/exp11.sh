#!/bin/bash
B="timeout 100 python bench.py --cpu-seconds 0 --steps 10 --warmup 2 --width 320 --height 240 --iters 50 --batch 1024"
short() { python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-52s %9.0f aligns/s  kernel %.3f ms' % (sys.argv[1], d['value'], d['roofline']['kernel_ms']))" "$1"; }
$B 2>/dev/null | short "refdefault 320x240x4x50: block=512 lds=auto"
$B --block 256 --lds-point-bytes 48000 2>/dev/null | short "refdefault: block=256 lds=48000 (2-3 WG/CU)"
$B --block 256 --lds-point-bytes 77000 2>/dev/null | short "refdefault: block=256 lds=77000 (2 WG/CU)"
$B --block 512 --lds-point-bytes 48000 2>/dev/null | short "refdefault: block=512 lds=48000"
B2="timeout 100 python bench.py --cpu-seconds 0 --steps 10 --warmup 2"
for i in 1 2 3; do $B2 2>/dev/null | short "C2 default run $i"; done
