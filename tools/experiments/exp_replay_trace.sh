#!/bin/bash
# kernel traces of the C++ replay as consecutive processes: in a slow process, are the KERNELS slow or the gaps between them?
R=$PWD; O=$R/gpurun_out/r03_single_stream; mkdir -p $O
FRAMES_DIR=/tmp/frames ONLY_GENERATE=1 python tests/tools/track_latency.py 640 480 4 10 16 > /dev/null
DEMO="$R/rgbd_odometry_amd/lib/track_demo /tmp/frames 0 15 1 4 525.0 525.0 319.5 239.5 10 /tmp/poses.txt"
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3 4 5; do
  rm -rf $O/replay_trace_$i
  TRACK_DEMO_VERBOSE=1 timeout 120 rocprofv3 --kernel-trace --output-format csv -d $O/replay_trace_$i -o t -- $DEMO 2>/dev/null | python3 -c "
import sys,re
t=[float(re.search(r': ([0-9.]+) ms',l).group(1)) for l in sys.stdin if l.startswith('frame ')]
print('process $i: frames %d  median %.3f ms  max %.3f ms' % (len(t), sorted(t)[len(t)//2], max(t)))"
  python3 - $O/replay_trace_$i <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
last_end = int(rows[0]["Start_Timestamp"])
gaps, durs, big = [], {}, []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    g = (s - last_end) / 1e3
    n = r["Kernel_Name"].split("(")[0].replace("void dvo::", "").replace("dvo::", "")[:28]
    durs.setdefault(n, []).append((e - s) / 1e3)
    if 300 < g < 11000: big.append((round(g), n))
    last_end = max(last_end, e)
print("   kernels %d; gaps of 0.3-11 ms before a kernel: %d  e.g. %s" % (len(rows), len(big), big[:6]))
print("   max kernel durations (us):", {k: round(max(v), 1) for k, v in sorted(durs.items(), key=lambda kv: -max(kv[1]))[:5]})
PY
done
