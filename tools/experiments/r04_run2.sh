#!/bin/bash
python -m pytest tests/test_gpu_capacity.py -m gpu -x -q > gpurun_out/r04_cap.txt 2>&1; grep -E "passed|failed|skipped" gpurun_out/r04_cap.txt | tail -2; grep -E "^E " gpurun_out/r04_cap.txt | head -10
python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_capacity.py > gpurun_out/r04_gputest3.txt 2>&1; grep -E "passed|failed" gpurun_out/r04_gputest3.txt | tail -2
for b in 16384 32768 40000; do
python bench.py --steps 20 --warmup 3 --cpu-seconds 0 --no-extra-legs --batch $b 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('batch $b: %.0f aligns/s frac %.4f  ms/step %.2f timed %.2f s' % (d['value'], d['roofline']['frac'], d['ms_per_step'], d['config']['timed_region_s']))"
done
