#!/bin/bash
# final record of the round: PMC traffic (ties to the kernel hash), the default bench line, the secondary configurations
python3 tools/update_pmc_traffic.py r03 > /dev/null 2>&1
python3 tools/update_pmc_traffic.py r03 --batch 1024 > /dev/null 2>&1
python3 tools/update_pmc_traffic.py r03 --width 1920 --height 1080 --levels 5 --batch 1024 --distinct 8 > /dev/null 2>&1
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
mkdir -p gpurun_out/r03_final
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_final/bench_driver_flags.json 2> gpurun_out/r03_final/bench.err; echo "bench rc=$?"
python bench.py > gpurun_out/r03_final/bench_defaults.json 2>> gpurun_out/r03_final/bench.err; echo "bench rc=$?"
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-extra-legs 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('repeat', round(d['value']), round(d['roofline']['frac'],4), round(d['roofline']['kernel_ms'],3), d['roofline']['traffic'])"; done | tee gpurun_out/r03_final/repeat_runs.txt
tools/other_configs.sh > gpurun_out/r03_final/other_configs_summary.txt 2>&1; cp gpurun_out/other_configs.txt gpurun_out/r03_final/
tail -30 gpurun_out/r03_final/other_configs_summary.txt
cut -c1-1200 gpurun_out/r03_final/bench_driver_flags.json
