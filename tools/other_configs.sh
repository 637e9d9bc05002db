#!/bin/bash
# raw result lines of the secondary configurations quoted in DESIGN.md section 6 -> gpurun_out/other_configs.txt
O=gpurun_out/other_configs.txt; mkdir -p gpurun_out; : > $O
run() { echo "### $*" >> $O; "$@" 2>&1 | grep -v "amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl path" | tail -${TAILN:-1} >> $O; }
run python bench.py --cpu-seconds 0 --no-frames-leg --width 1920 --height 1080 --levels 5 --batch 512 --distinct 8 --steps 5 --warmup 1
run python bench.py --cpu-seconds 0 --no-frames-leg --width 320 --height 240 --iters 50
run python bench.py --cpu-seconds 0 --no-frames-leg --batch 256
run python bench.py --cpu-seconds 0 --no-frames-leg --batch 4096
run python tools/bench_tiled.py --width 4096 --height 3072 --levels 5 --steps 20
run python tools/bench_tiled.py --width 1920 --height 1080 --levels 5 --steps 20
run python tools/bench_tiled.py --width 640 --height 480 --levels 4 --steps 20
TAILN=3 run python tools/track_latency.py 320 240 4 50 16
TAILN=3 run python tools/track_latency.py 640 480 4 10 16
cat $O | cut -c1-420
