#!/bin/bash
# the round-3 rocprofv3 record: kernel trace + PMC passes of every configuration DESIGN.md section 6 quotes
tools/profile.sh r03_c2_b16384 > /dev/null 2>&1
tools/profile.sh r03_c2_b8192 --batch 8192 > /dev/null 2>&1
tools/profile.sh r03_c2_b1024 --batch 1024 > /dev/null 2>&1
tools/profile.sh r03_c2_b1024_texels16 --batch 1024 --variant 4 > /dev/null 2>&1
tools/profile.sh r03_c2_b32_teams --batch 32 > /dev/null 2>&1
tools/profile.sh r03_c3_b256 --width 1920 --height 1080 --levels 5 --batch 256 --distinct 8 > /dev/null 2>&1
tools/profile.sh r03_c3_b1024 --width 1920 --height 1080 --levels 5 --batch 1024 --distinct 8 > /dev/null 2>&1
tools/profile.sh r03_c3_b2048 --width 1920 --height 1080 --levels 5 --batch 2048 --distinct 8 > /dev/null 2>&1
for t in r03_c2_b16384 r03_c2_b8192 r03_c2_b1024 r03_c2_b1024_texels16 r03_c2_b32_teams r03_c3_b256 r03_c3_b1024 r03_c3_b2048; do echo "== $t"; grep "align_fused" gpurun_out/prof_$t/summary.txt | head -24; tail -1 gpurun_out/prof_$t/trace.log | cut -c1-300; done
python3 tools/update_pmc_traffic.py r03 > /dev/null 2>&1
python3 tools/update_pmc_traffic.py r03 --batch 8192 > /dev/null 2>&1
python3 tools/update_pmc_traffic.py r03 --batch 1024 > /dev/null 2>&1; 
python3 tools/update_pmc_traffic.py r03 --width 1920 --height 1080 --levels 5 --batch 256 --distinct 8 > /dev/null 2>&1
python3 tools/update_pmc_traffic.py r03 --width 1920 --height 1080 --levels 5 --batch 1024 --distinct 8 > /dev/null 2>&1
python3 tools/update_pmc_traffic.py r03 --width 1920 --height 1080 --levels 5 --batch 2048 --distinct 8 > /dev/null 2>&1
cat gpurun_out/pmc_traffic.json
# (the frame path's record: tools/r03_frames_record.sh)
