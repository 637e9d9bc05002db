#!/bin/bash
# the C++ file replay (examples/track_demo.cpp: ~17 ms of host work between frames) as first and later process of a box,
# without and with the keep-warm thread (DVO_KEEP_WARM="busy_us,pause_us")
O=gpurun_out/r03_single_stream; mkdir -p $O
run() { echo "== $1 (DVO_KEEP_WARM='$2')"; DVO_KEEP_WARM="$2" VERBOSE=1 python tests/tools/track_latency.py 640 480 4 10 16 2>&1 | grep "frame \|per frame" | awk 'NR<=2 || NR>=14' | cut -c1-120; }
{
run "first process of the box" ""
run "second process" ""
run "third process, keep-warm one wave busy all the time" "200,0"
run "keep-warm 10 % duty" "200,1800"
run "keep-warm 1 % duty" "50,5000"
run "no keep-warm again" ""
} 2>&1 | tee $O/cpp_replay_keep_warm.txt
