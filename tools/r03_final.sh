#!/bin/bash
# final record of the round: PMC traffic (ties to the kernel hash), the default bench line, the secondary configurations
# (PMC traffic: tools/r03_profiles.sh)
mkdir -p gpurun_out/r03_final
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_final/bench_driver_flags.json 2> gpurun_out/r03_final/bench.err; echo "bench rc=$?"
python bench.py > gpurun_out/r03_final/bench_defaults.json 2>> gpurun_out/r03_final/bench.err; echo "bench rc=$?"
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-extra-legs 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('repeat', round(d['value']), round(d['roofline']['frac'],4), round(d['roofline']['kernel_ms'],3), d['roofline']['traffic'])"; done | tee gpurun_out/r03_final/repeat_runs.txt
tools/other_configs.sh > gpurun_out/r03_final/other_configs_summary.txt 2>&1; cp gpurun_out/other_configs.txt gpurun_out/r03_final/
tail -30 gpurun_out/r03_final/other_configs_summary.txt
cut -c1-1200 gpurun_out/r03_final/bench_driver_flags.json
