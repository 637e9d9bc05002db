#!/bin/bash
# the round-6 record on the GPU box: kernel trace + PMC passes per configuration, PMC traffic keyed to the kernel hash, the bench
# lines (driver flags / defaults), secondary configurations.  usage: tools/r06_record.sh [part ...]   parts: prof traffic bench other frames tiled
PARTS=${*:-"traffic prof bench other frames tiled extra"}
has() { case " $PARTS " in *" $1 "*) return 0;; esac; return 1; }
mkdir -p gpurun_out/r06_final
if has traffic; then
  python3 tools/update_pmc_traffic.py r06 > /dev/null 2>&1
  python3 tools/update_pmc_traffic.py r06 --batch 1024 > /dev/null 2>&1
  python3 tools/update_pmc_traffic.py r06 --width 1920 --height 1080 --levels 5 --batch 1024 --distinct 8 > /dev/null 2>&1
  cat gpurun_out/pmc_traffic.json
  [ -f gpurun_out/pmc_traffic.json ] && cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
fi
if has prof; then
  tools/profile.sh r06_c2_b40000 > /dev/null 2>&1
  tools/profile.sh r06_c2_b1024 --batch 1024 > /dev/null 2>&1
  tools/profile.sh r06_c2_b32_teams --batch 32 > gpurun_out/r06_final/profile_teams.log 2>&1; echo "teams profile rc=$?"
  tools/profile.sh r06_c3_b1024 --width 1920 --height 1080 --levels 5 --batch 1024 --distinct 8 > /dev/null 2>&1
  for t in r06_c2_b40000 r06_c2_b1024 r06_c2_b32_teams r06_c3_b1024; do echo "== $t"; grep "align_fused" gpurun_out/prof_$t/summary.txt | head -24; cut -c1-200 gpurun_out/prof_$t/bench_line_profiled.json; echo; grep -c "SIGSEGV" gpurun_out/prof_$t/*.log | tr '\n' ' '; echo; done
fi
if has bench; then
  [ -f gpurun_out/pmc_traffic.json ] && cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
  python bench.py --steps 20 --warmup 5 > gpurun_out/r06_final/bench_driver_flags.json 2> gpurun_out/r06_final/bench.err; echo "bench rc=$?"
  for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-extra-legs 2>/dev/null | python3 -c "
import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('repeat', round(d['value']), round(d['roofline']['frac'],4), round(d['roofline']['kernel_ms'],3), d['roofline']['traffic'])"; done | tee gpurun_out/r06_final/repeat_runs.txt
  cut -c1-1500 gpurun_out/r06_final/bench_driver_flags.json
fi
if has other; then
  tools/other_configs.sh > gpurun_out/r06_final/other_configs_summary.txt 2>&1; cp gpurun_out/other_configs.txt gpurun_out/r06_final/ 2>/dev/null
  tail -30 gpurun_out/r06_final/other_configs_summary.txt
fi
if has tiled; then
  for a in "--width 4096 --height 3072 --levels 5" "--width 1920 --height 1080 --levels 5" "--width 640 --height 480 --levels 4"; do
    python bench.py --mode tiled --cpu-seconds 2 $a 2>/dev/null | grep '^{' >> gpurun_out/r06_final/tiled_lines.json
  done
  python3 -c "
import json
for l in open('gpurun_out/r06_final/tiled_lines.json'):
    d=json.loads(l); print(d['config']['workload'][:40], '%.1f aligns/s %.3f ms %.2f us/iter step %.2f us graph=%s parity=%s solo=%s N=%s' % (d['value'], d['ms_per_step'], d['config']['us_per_iteration'], 1e3*d['roofline']['kernel_ms'], d['config']['graph_replayed'], d.get('parity_check',{}).get('pass'), d['config'].get('levels_as_one_launch'), d['config'].get('points_per_level')))"
fi
if has frames; then
  python tools/bench_frames.py --batch 256 --pinned --reps 10 > gpurun_out/r06_final/bench_frames_640x480_b256_pinned.json 2>/dev/null
  python tools/single_stream.py > gpurun_out/r06_final/single_stream.txt 2>&1; tail -5 gpurun_out/r06_final/single_stream.txt
fi
if has extra; then
  # round 6: stamps anatomy (serial chain), sparse scenes, H on the packed kernel, the team exchange's store flavour
  # the stamps build travels with the tree: make -C rgbd_odometry_amd/csrc STAMPS=1 (before gpurun)
  if [ -f rgbd_odometry_amd/lib/libdvo_amd_stamps.so ]; then
    for cfg in "2048 0" "256 0" "32 0"; do echo "== stamps B/block = $cfg"; PREP=1 python tools/experiments/exp_stamps2.py $cfg 2>&1 | tail -6; done > gpurun_out/r06_final/stamps_anatomy.txt 2>&1
  else echo "stamps anatomy skipped: libdvo_amd_stamps.so not built"; fi
  python tools/sparse_scenes.py > gpurun_out/r06_final/sparse_scenes.json 2> gpurun_out/r06_final/sparse_scenes.txt
  for b in 1024 8192; do python bench.py --cpu-seconds 0 --no-extra-legs --batch $b --steps 20 --normal-matrix 2>/dev/null | grep '^{'; done > gpurun_out/r06_final/normal_matrix_lines.json
  TEAMS=64,128,256,0 python tools/experiments/exp_team_single.py 4096 3072 5 2>&1 | tail -1 > gpurun_out/r06_final/team_single_4096.txt
  tail -3 gpurun_out/r06_final/stamps_anatomy.txt; grep "^#" gpurun_out/r06_final/sparse_scenes.txt | head -3; cat gpurun_out/r06_final/team_single_4096.txt
fi
