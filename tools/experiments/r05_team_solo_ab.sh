#!/bin/bash
# round 5: solo levels of the team kernel (dvo_fused.hip) -- a level with at most DVO_TEAM_SOLO_MAX points is run by member 0 alone.
# One pair, ms per alignment (20 back to back) by threshold, three frame sizes; then small batches through bench.py.
# usage (GPU box): tools/experiments/r05_team_solo_ab.sh      output: stdout
for thr in 0 1024 2048 4096 8192 16384 65536; do
  echo "== DVO_TEAM_SOLO_MAX=$thr"
  for cfg in "640 480 4" "1920 1080 5" "4096 3072 5"; do
    DVO_TEAM_SOLO_MAX=$thr TEAMS=0 python tools/experiments/exp_team_single.py $cfg 2>&1 | tail -1
  done
  for b in 8 32; do
    DVO_TEAM_SOLO_MAX=$thr python bench.py --cpu-seconds 0 --no-extra-legs --batch $b --steps 50 --warmup 5 2>/dev/null | grep '^{' | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('   batch $b: %.0f aligns/s  %.3f ms per step' % (d['value'], d['ms_per_step']))"
  done
done
