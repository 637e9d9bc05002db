#!/bin/bash
B="timeout 200 python bench.py --cpu-seconds 0 --steps 10 --warmup 2"
short() { python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-40s %9.0f aligns/s  kernel %.3f ms  frac %.3f' % (sys.argv[1], d['value'], d['roofline']['kernel_ms'], d['roofline']['frac']))" "$1"; }
for b in 256 512 1024 2048 4096 8192; do $B --batch $b 2>/dev/null | short "batch=$b"; done
