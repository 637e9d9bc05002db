#!/bin/bash
# round 6: launch sequence of ONE large pair through dvo_align_batch (tiers of team launches vs one team size) and the wide path.
# usage (GPU box): tools/experiments/r06_team_single_trace.sh W H levels "teams"      output: gpurun_out/r06_team_single_trace/
W=${1:-4096}; H=${2:-3072}; L=${3:-5}; TE=${4:-"0 256"}
OUT=$PWD/gpurun_out/r06_team_single_trace; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD
for t in $TE; do
  cd /tmp; rm -rf $OUT/trace
  PYTHONPATH=$REPO TEAMS=$t rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o trace -- python3 $REPO/tools/experiments/exp_team_single.py $W $H $L > $OUT/trace_$t.log 2>&1
  cd $REPO
  echo "== team $t: $(tail -1 $OUT/trace_$t.log | cut -c1-200)"
  python3 - <<PY
import csv, glob
f = glob.glob('$OUT/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
tail = rows[-${TAILN:-24}:]
prev = None
for r in tail:
    n = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('dvo::', '')[:44]
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%-44s grid %-8s %.1f us (+%.1f)' % (n, r.get('Grid_Size_X', r.get('Grid_Size', '?')), (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0))
    prev = e
PY
done
rm -rf $OUT/trace
