#!/bin/bash
# round 5: the distance transform's row pass, four rows per lane (ds_read_b64) against two (ds_read_b32): kernel durations of the
# now-frame stage over 256 camera frames (rocprofv3 kernel trace of tools/bench_frames.py).  Variants are libraries built with
#   make -C rgbd_odometry_amd/csrc EXP=rowsb32 EXPDEFS=-DDVO_EDT_ROWS_B32=1      (and whatever else the caller built)
# usage (GPU box): tools/experiments/r05_edt_rows_ab.sh [variant ...]     ("" = the default library)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in "" "$@"; do
  [ -n "$v" ] && [ ! -f $R/rgbd_odometry_amd/lib/libdvo_amd_$v.so ] && continue
  O=$R/gpurun_out/edt_rows_ab/${v:-default}; mkdir -p $O; rm -rf $O/trace
  DVO_LIB_VARIANT=${v:+_$v} timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o frames -- python3 $R/tools/bench_frames.py --batch 256 --pinned --reps 3 > $O/trace.log 2>&1
  echo "== variant ${v:-default}"
  python3 - <<PY
import csv, glob
f = glob.glob('$O/trace/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r['Name']
    if 'edt_' in n or 'dt_normalize' in n:
        print('   %-52s calls %4s  avg %9.1f us' % (n.split('(')[0].replace('void ', '')[:52], r['Calls'], float(r['AverageNs']) / 1e3))
PY
  grep -o '"as_now_ms_per_frame[^,]*' $O/trace.log | head -2
  rm -rf $O/trace
done
