#!/bin/bash
# round 6: kernel timeline of one `camera frames in HBM -> poses` step (256 four-level 640x480 frames).  output: gpurun_out/r06_device_step/
OUT=$PWD/gpurun_out/r06_device_step; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD
python3 tools/experiments/r06_device_step.py 20 2>&1 | grep device_step
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python3 $REPO/tools/experiments/r06_device_step.py 6 > $OUT/trace.log 2>&1
cd $REPO
python3 - <<PY
import csv, glob
f = glob.glob('$OUT/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
al = [i for i, r in enumerate(rows) if 'align_fused2' in r['Kernel_Name']]
a, b = al[-2], al[-1]
win = rows[a + 1:b + 1]
t0 = int(rows[a]['End_Timestamp'])
span = (int(rows[b]['End_Timestamp']) - t0) / 1e3
print('one step: %.1f us from the end of the previous alignment to the end of this one, %d launches' % (span, len(win)))
prev = t0
tot = {}
for r in win:
    n = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('dvo::', '')[:44]
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%-44s %8.1f us  (+%.1f)' % (n, (e - s) / 1e3, (s - prev) / 1e3))
    tot[n] = tot.get(n, 0) + (e - s) / 1e3
    prev = max(prev, e)
print('busy %.1f us of %.1f' % (sum(tot.values()), span))
PY
rm -rf $OUT/trace
