#!/bin/bash
# record of the frame path (rows f1 + f2) for profiles/r03_frames: stage timings, kernel trace, PMC passes, GPU time per now frame
# against the round-2 tree (if baseline_r2/ is there), single camera stream
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03_frames_record; mkdir -p $O
tools/prof_frames.sh > $O/prof.log 2>&1; cp gpurun_out/prof_frames/summary.txt $O/summary.txt; cp gpurun_out/prof_frames/pinned.json $O/bench_frames_640x480_b256_pinned.json
cp gpurun_out/prof_frames/trace/frames_kernel_stats.csv $O/kernel_stats.csv
bash tools/pmc_frames.sh > $O/pmc.log 2>&1; cp gpurun_out/pmc_frames/summary.txt $O/pmc_summary.txt
[ -d $R/baseline_r2 ] && bash tools/experiments/exp_now_frame_gpu_time.sh 2>&1 | grep -v amdgpu.ids > $O/gpu_time_per_now_frame_r2_vs_r3.txt
python3 tools/single_stream.py --out $O/single_stream.json > $O/single_stream.txt 2>&1
tail -5 $O/single_stream.txt; wc -l $O/*
