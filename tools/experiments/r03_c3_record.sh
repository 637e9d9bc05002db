tools/profile.sh r03_c3_b1024 --width 1920 --height 1080 --levels 5 --batch 1024 --distinct 8 > /dev/null 2>&1
tools/profile.sh r03_c3_b2048 --width 1920 --height 1080 --levels 5 --batch 2048 --distinct 8 > /dev/null 2>&1
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json
python3 tools/update_pmc_traffic.py r03 --width 1920 --height 1080 --levels 5 --batch 1024 --distinct 8 > /dev/null 2>&1
python3 tools/update_pmc_traffic.py r03 --width 1920 --height 1080 --levels 5 --batch 2048 --distinct 8 > /dev/null 2>&1
for t in r03_c3_b1024 r03_c3_b2048; do echo "== $t"; grep "align_fused" gpurun_out/prof_$t/summary.txt | head -3 | cut -c1-220; done
python3 -c "
import json; d=json.load(open('gpurun_out/pmc_traffic.json')); print({k:(v['hbm_bytes_per_launch'], v['l2_read_requests']) for k,v in d.items()})"
tools/other_configs.sh > gpurun_out/other_configs_summary.txt 2>&1; head -8 gpurun_out/other_configs_summary.txt | cut -c1-200
