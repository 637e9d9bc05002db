#!/bin/bash
# the default configuration (16384 pairs per launch): kernel trace + PMC, traffic record, the bench lines
tools/profile.sh r03_c2_b16384 > /dev/null 2>&1
grep "align_fused" gpurun_out/prof_r03_c2_b16384/summary.txt | head -3 | cut -c1-220
python3 tools/update_pmc_traffic.py r03 > /dev/null 2>&1
python3 -c "
import json; d=json.load(open('gpurun_out/pmc_traffic.json')); print({k:(v['hbm_bytes_per_launch'], v['l2_read_requests'], v['kernel_source_sha256']) for k,v in d.items()})"
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
bash tools/r03_final.sh 2>&1 | tail -12
