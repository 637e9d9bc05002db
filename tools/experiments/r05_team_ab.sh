#!/bin/bash
# round 5: team records as plain stores inside a verified XCD (default) vs always sc1 (DVO_TEAM_PLAIN_STORES=off), same box, interleaved
run() { python bench.py --no-extra-legs --cpu-seconds 0 "$@" 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%9.1f aligns/s  %.4f ms/step' % (d['value'], d['ms_per_step']))"; }
for rep in 1 2 3; do for mode in plain off; do
  for b in 32 64 128; do echo -n "$mode b$b : "; DVO_TEAM_PLAIN_STORES=$mode run --batch $b --steps 400 --warmup 20; done
done; done
for mode in plain off; do echo "== single pair 4096x3072 teams ($mode)"; DVO_TEAM_PLAIN_STORES=$mode TEAMS=128,0 python tools/experiments/exp_team_single.py 4096 3072 5 2>&1 | tail -1; done
