#!/bin/bash
# round 5: parity sweep through the bench's own in-run check -- every pair of the batch a DIFFERENT synthetic scene, each aligned by the fused
# kernel on natively produced compact now levels and by the CPU oracle; energies / best index / visible ratio compared bit for bit.
# usage (GPU box): tools/experiments/r05_parity_sweep.sh > gpurun_out/r05_final/parity_sweep.txt
run() {   # label, N, extra flags
  echo "$1, $2 DISTINCT scenes in one batch of $2"
  python bench.py --distinct $2 --batch $2 --no-extra-legs --cpu-seconds 2 --no-cpu-all-cores --steps 5 --warmup 1 $3 2>/dev/null | grep '^{' | python3 -c "
import json, sys
d = json.loads(sys.stdin.read())
print('  parity_check:', json.dumps(d.get('parity_check')))
print('  throughput of that batch: %.0f aligns/s, kernel %.3f ms, frac %.4f' % (d['value'], d['roofline']['kernel_ms'], d['roofline']['frac']))"
}
run "C2 640x480x4x10" 2048 ""
run "C2 640x480x4x10, DVO_FLAG_NORMAL_MATRIX" 512 "--normal-matrix"
run "C3 1920x1080x5x10" 128 "--width 1920 --height 1080 --levels 5"
run "320x240x4x50 (the reference's default schedule)" 512 "--width 320 --height 240 --iters 50"
for cfg in "--width 4096 --height 3072 --levels 5" "--width 1920 --height 1080 --levels 5" "--width 640 --height 480 --levels 4"; do
  echo "tiled / wide schedule, one pair, $cfg"
  python bench.py --mode tiled --cpu-seconds 1 $cfg 2>/dev/null | grep '^{' | python3 -c "
import json, sys
d = json.loads(sys.stdin.read())
print('  parity_check:', json.dumps(d.get('parity_check')), ' %.3f ms per alignment' % d['ms_per_step'])"
done
