#!/bin/bash

for rep in 1 2; do
for v in "DVO_EDT_SPLIT=1" "DVO_EDT_SPLIT=0"; do
  env $v python tools/bench_frames.py --batch 256 --pinned --reps 10 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v', {k[:30]: round(v['ms'],3) for k,v in d['stages'].items() if isinstance(v,dict) and 'ms' in v and ('as_now' in k or 'align' in k or 'Canny' in k or 'as_ref' in k)})"
done
done
