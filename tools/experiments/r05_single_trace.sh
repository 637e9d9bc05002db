#!/bin/bash
# round 5: GPU anatomy of ONE camera frame through the single-stream path (kernel trace of tools/single_stream.py, back to back):
# per kernel the calls per frame and the time per frame, and the idle time between consecutive launches of one frame.
# usage (GPU box): tools/experiments/r05_single_trace.sh      output: gpurun_out/r05_single_trace/summary.txt
OUT=$PWD/gpurun_out/r05_single_trace; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD
cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace -o t -- python3 $REPO/tools/single_stream.py --frames 100 --order back_to_back > $OUT/run.log 2>&1
cd $REPO
python3 - <<PY | tee $OUT/summary.txt
import csv, glob
f = glob.glob('$OUT/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# the last 60 frames: a frame starts at its first camera-stage kernel; find the per-frame period by the alignment kernel
al = [i for i, r in enumerate(rows) if 'align_fused2' in r['Kernel_Name']]
al = al[-61:]
seg = rows[al[0] + 1: al[-1] + 1]          # 60 whole frames, each ending with its alignment kernel
nfr = 60
acc = {}
for r in seg:
    n = r['Kernel_Name'].split('(')[0].replace('void ', '')[:58]
    acc.setdefault(n, []).append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
tot = 0
for n, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    tot += sum(v)
    print('%-58s %5.2f calls/frame %8.2f us/frame' % (n, len(v) / nfr, sum(v) / 1e3 / nfr))
span = (int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e3 / nfr
print('kernel time %.1f us per frame, %d launches per frame; frame period on the GPU %.1f us' % (tot / 1e3 / nfr, len(seg) / nfr, span))
gaps = {}
for a, b in zip(seg, seg[1:]):
    g = (int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3
    k = a['Kernel_Name'].split('(')[0].replace('void ', '')[:40] + ' -> ' + b['Kernel_Name'].split('(')[0].replace('void ', '')[:40]
    gaps.setdefault(k, []).append(g)
print('idle between consecutive kernels (median us):')
import statistics
for k, v in sorted(gaps.items(), key=lambda kv: -statistics.median(kv[1]) * len(kv[1]))[:16]:
    print('   %-84s n/frame %.2f median %7.2f' % (k, len(v) / nfr, statistics.median(v)))
PY
tail -3 $OUT/run.log
rm -rf $OUT/trace
