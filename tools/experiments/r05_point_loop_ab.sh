#!/bin/bash
# round 5: the two "last levers" on the point loop of the throughput kernel, on real data, same box, interleaved:
#   new      the product library
#   diet     lane masks in scalar registers + clamp-free full rounds (make EXP=diet EXPDEFS=-DDVO_VALU_DIET=1)
#   r16      ranks of the coarse levels (2, 3 at 640x480) from an LDS copy of the level (make EXP=r16 EXPDEFS=-DDVO_ENABLE_R16=1)
#   dietr16  both
run() { lib=""; [ "$1" != "new" ] && lib="_$1"; shift
  DVO_LIB_VARIANT=$lib python bench.py --no-extra-legs --cpu-seconds 0 "$@" 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%9.1f aligns/s  kernel %.3f ms  frac %.4f  parity %s' % (d['value'], d['roofline'].get('kernel_ms') or 0, d['roofline']['frac'], (d.get('parity_check') or {}).get('pass')))"; }
for rep in 1 2; do for v in new diet r16 dietr16; do
  echo -n "$v c2 b8192 : "; run $v --batch 8192 --steps 30
  echo -n "$v c2 b1024 : "; run $v --batch 1024 --steps 100
  echo -n "$v c3 b1024 : "; run $v --width 1920 --height 1080 --levels 5 --batch 1024 --distinct 8 --steps 5 --warmup 1
done; done
for v in diet dietr16; do echo "== parity of $v (bench's own check, 32 scenes)"; DVO_LIB_VARIANT=_$v python bench.py --no-extra-legs --cpu-seconds 2 --no-cpu-all-cores --batch 1024 --steps 5 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d.get('parity_check'))"; done
