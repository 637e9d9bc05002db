#!/bin/bash
O=gpurun_out/r03_single_stream; mkdir -p $O
run() { echo "== $1 ($2)"; env $2 VERBOSE=1 python tests/tools/track_latency.py 640 480 4 10 16 2>&1 | grep "frame \|per frame" | awk 'NR<=2 || NR>=15' | cut -c1-120; }
{
run "first process of the box" "X=1"
run "second process" "X=1"
run "third process, polling waits" "HSA_ENABLE_INTERRUPT=0"
run "fourth, default again" "X=1"
run "fifth, blocking sync flag" "HIP_FORCE_BLOCKING_SYNC=1"
run "sixth, polling waits again" "HSA_ENABLE_INTERRUPT=0"
} 2>&1 | tee $O/cpp_replay_wait_mode.txt
rocm-smi --showclocks 2>/dev/null | head -20
