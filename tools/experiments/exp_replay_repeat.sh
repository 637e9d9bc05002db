#!/bin/bash
# the C++ file replay as 6 consecutive processes of one box: worst and mean frame per process
FRAMES_DIR=/tmp/frames ONLY_GENERATE=1 python tests/tools/track_latency.py 640 480 4 10 16 > /dev/null
DEMO="rgbd_odometry_amd/lib/track_demo /tmp/frames 0 15 1 4 525.0 525.0 319.5 239.5 10 /tmp/poses.txt"
for i in 1 2 3 4 5 6; do
  env $1 TRACK_DEMO_VERBOSE=1 $DEMO | python3 -c "
import sys,re
t=[float(re.search(r': ([0-9.]+) ms',l).group(1)) for l in sys.stdin if l.startswith('frame ')]
print('process $i ($1): frames %d  median %.3f ms  max %.3f ms  mean %.3f ms' % (len(t), sorted(t)[len(t)//2], max(t), sum(t)/len(t)))"
  sleep 1
done
