#!/bin/bash
# the committed single-stream record (profiles/r03_single_stream/): Python-driven stream at 30 Hz + the C++ file replay as
# consecutive processes of one box
O=gpurun_out/r03_single_stream; mkdir -p $O
python tools/single_stream.py --frames 200 --out $O/summary.json 2>&1 | grep "host wall" | cut -c1-400 | tee $O/python_stream.txt
{ echo "# C++ file replay (examples/track_demo.cpp), 16 frames of 640x480x4x10, six consecutive processes"; tools/experiments/exp_replay_repeat.sh FINAL=1; } | tee $O/cpp_replay_repeat.txt
