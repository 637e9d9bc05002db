#!/bin/bash
# session driver: tests, tiled bench lines, default bench line (short)
python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputest2.txt 2>&1; grep -E "passed|failed" gpurun_out/r04_gputest2.txt | tail -2
for a in "" "DVO_TILED_NO_GRAPH=1"; do
  env $a python bench.py --mode tiled --cpu-seconds 1 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('tiled 4096x3072 [$a]: %.1f aligns/s  %.3f ms/align  %.2f us/iter  step %.2f us  graph=%s parity=%s' % (d['value'], d['ms_per_step'], d['config']['us_per_iteration'], 1e3*d['roofline']['kernel_ms'], d['config'].get('graph_replayed'), d.get('parity_check',{}).get('pass')))"
done
python bench.py --mode tiled --width 640 --height 480 --levels 4 --cpu-seconds 0 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('tiled 640x480: %.1f aligns/s  %.2f us/iter graph=%s' % (d['value'], d['config']['us_per_iteration'], d['config'].get('graph_replayed')))"
python tools/bench_tiled.py 2>&1 | tail -6
python bench.py --steps 20 --warmup 3 --cpu-seconds 2 > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err; python -c "
import json
d=json.loads([l for l in open('gpurun_out/r04_bench_default.json') if l.startswith('{')][-1])
print('default: %.0f aligns/s frac %.4f parity %s' % (d['value'], d['roofline']['frac'], d.get('parity_check',{}).get('pass')))
print('float_now_levels:', json.dumps(d.get('float_now_levels'))[:900])
"
