#!/bin/bash
# round 5: where a tiled iteration's time goes -- kernel durations (rocprofv3 kernel trace) next to the replayed graph's wall time.
# usage (GPU box): tools/experiments/r05_tiled_trace.sh [bench args]     output: gpurun_out/r05_tiled_trace/
OUT=$PWD/gpurun_out/r05_tiled_trace; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD
ARGS=${*:-"--width 640 --height 480 --levels 4"}
python bench.py --mode tiled --cpu-seconds 0 $ARGS 2>/dev/null | grep '^{' > $OUT/line.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $REPO/bench.py --mode tiled --cpu-seconds 0 --steps 20 --warmup 2 $ARGS > $OUT/trace.log 2>&1
cd $REPO
python3 - <<PY
import csv, glob, json
d = json.loads(open('$OUT/line.json').read())
print('bench: %.3f ms per alignment, %.2f us per iteration' % (d['ms_per_step'], d['config']['us_per_iteration']))
f = glob.glob('$OUT/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = {}
for r in rows:
    n = r['Kernel_Name'].split('(')[0][:60]
    names.setdefault(n, []).append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for n, v in sorted(names.items(), key=lambda kv: -sum(kv[1]))[:8]:
    print('%-60s n=%6d avg %.2f us  min %.2f' % (n, len(v), sum(v) / len(v) / 1e3, min(v) / 1e3))
# gaps between consecutive tiled_step launches (end -> next start) over the last alignment
st = [r for r in rows if 'tiled_step' in r['Kernel_Name']]
tail = st[-40:]
gaps = [(int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3 for a, b in zip(tail, tail[1:])]
durs = [(int(a['End_Timestamp']) - int(a['Start_Timestamp'])) / 1e3 for a in tail]
print('last 40 step launches: durations', ' '.join('%.1f' % x for x in durs))
print('gaps to the next launch      ', ' '.join('%.1f' % x for x in gaps))
PY
python3 tools/experiments/r05_trace_sequence.py $OUT/trace > $OUT/sequence.txt 2>&1; rm -rf $OUT/trace
