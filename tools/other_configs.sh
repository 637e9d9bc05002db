#!/bin/bash
# raw result lines of the secondary configurations quoted in DESIGN.md section 6 -> gpurun_out/other_configs.txt
O=gpurun_out/other_configs.txt; mkdir -p gpurun_out; : > $O
run() { echo "### $*" >> $O; "$@" 2>&1 | grep -v "amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl path" | tail -${TAILN:-1} >> $O; }
Q="--cpu-seconds 0 --no-extra-legs"
run python bench.py $Q --width 1920 --height 1080 --levels 5 --batch 256 --distinct 8 --steps 5 --warmup 1
run python bench.py $Q --width 1920 --height 1080 --levels 5 --batch 1024 --distinct 8 --steps 5 --warmup 1
run python bench.py $Q --width 1920 --height 1080 --levels 5 --batch 2048 --distinct 8 --steps 4 --warmup 1
run python bench.py $Q --width 320 --height 240 --iters 50 --batch 1024 --steps 20
run python bench.py $Q --batch 256 --steps 20
run python bench.py $Q --batch 1024 --steps 20
run python bench.py $Q --batch 4096 --steps 20
run python bench.py $Q --batch 32 --steps 20
run python bench.py $Q --batch 64 --steps 20
run python bench.py $Q --batch 128 --steps 20
DVO_TEAM_COOP_LAUNCH=1 run python bench.py $Q --batch 32 --steps 20
run python bench.py $Q --batch 1024 --steps 20 --float-now-levels
run python bench.py $Q --batch 1024 --steps 20 --float-now-levels --prepare
run python bench.py --mode tiled --steps 50 --cpu-seconds 0
run python bench.py --mode tiled --steps 50 --cpu-seconds 0 --width 1920 --height 1080 --levels 5
python3 - <<'PY'
import json
for l in open("gpurun_out/other_configs.txt"):
    if l.startswith("###"): print(l.strip()[:150]); continue
    try:
        d = json.loads(l)
        print("    %.0f %s  ms/step %.3f  frac %.4f  kernel_ms %.4f  %s" % (d["value"], d["unit"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms"], d["roofline"]["kernel"][:60]))
    except Exception as e:
        print("    ??", l[:200])
PY
