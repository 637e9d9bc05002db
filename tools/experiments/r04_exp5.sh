#!/bin/bash
# byte-form proxy v2 (tools/experiments/r04_patches/b1_proxy_v2.patch, make EXP=b1v2 EXPDEFS=-DDVO_PROXY_B1V2=1): the REAL access
# pattern of a one-byte-per-pixel now form -- unaligned 16-byte window + the tile's 16-bit base from the same line, byte extraction,
# wide-tile test -- on wrong data.  Same-box A/B against the product library.
run() { v=$1; shift; lib=""; [ "$v" != "base" ] && lib="_$v"
  DVO_LIB_VARIANT=$lib python bench.py --no-extra-legs --cpu-seconds 0 "$@" 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('%9d aligns/s  kernel %.3f ms  frac %.4f' % (round(d['value']), r['kernel_ms'], r['frac']))"; }
# b1a = window rounded down to a dword (isolates the misalignment), b1n = no base load, b1an = both
for rep in 1 2; do for v in ${VARS:-base b1v2 b1a b1n b1an}; do
  echo -n "$v c2 b8192 : "; run $v --batch 8192 --steps 30
  echo -n "$v c3 b1024 : "; run $v --width 1920 --height 1080 --levels 5 --batch 1024 --distinct 8 --steps 5 --warmup 1
done; done
