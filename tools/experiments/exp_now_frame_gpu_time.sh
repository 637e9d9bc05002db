#!/bin/bash
# GPU kernel time per now frame (640x480, batch 256), this tree against the round-2 tree extracted to baseline_r2/
# (git archive 42cc2cb | tar -x -C baseline_r2; make -C baseline_r2/rgbd_odometry_amd/csrc)
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for tree in $R/baseline_r2 $R; do
  [ -d $tree/rgbd_odometry_amd/lib ] || continue
  rm -rf /tmp/nfprof
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/nfprof -o t -- python3 $R/tools/experiments/gpu_time_per_now_frame.py $tree 4 > /tmp/nf.log 2>&1
  echo "== $tree"; grep -E "frames|Error|error" /tmp/nf.log | head -3
  python3 - <<'PY'
import csv, glob
reps, B = 5, 256        # warm-up step + 4 timed: all five are in the trace; the reference upload + as_ref are too (once)
for f in glob.glob('/tmp/nfprof/**/*kernel_trace.csv', recursive=True):
    acc = {}
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].split('(')[0].replace('void ', '')
        acc.setdefault(n, []).append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    tot = 0.0
    for n, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        if n.startswith('__amd_rocclr'): continue
        if 'enlist' in n or 'points4' in n or 'import_' in n: continue          # reference side, once
        tot += sum(v)
        print('   %-46s calls %5d  total %9.1f us' % (n[:46], len(v), sum(v) / 1e3))
    # the reference frames' own camera_level + Canny (one batch of 256) are in the total: 6 batches of camera/Canny, 5 of the rest
    print('   kernel time, all now-frame kernels: %.1f us  (%.3f us per frame over %d now-frame batches + 1 reference batch of camera/Canny)' % (tot / 1e3, tot / 1e3 / (reps * B), reps))
PY
done
