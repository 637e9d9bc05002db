#!/bin/bash
# rows per workgroup of the distance transform's row pass (DVO_EDT_ROWS = 16 / 8 / 4): frames_as_now per 256 frames + kernel times
for r in 16 8 4; do
  echo "DVO_EDT_ROWS=$r"
  DVO_EDT_ROWS=$r python tools/bench_frames.py --batch 256 --reps 3 --pinned 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,v in d['stages'].items():
    if 'as_now' in k: print('%-70s %8.3f ms %10.0f /s'%(k,v['ms'],v['per_s']))
"
done
cd /tmp; export TMPDIR=/tmp
for r in 16 8; do
  rm -rf /tmp/edtprof; DVO_EDT_ROWS=$r timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/edtprof -o t -- python3 $GRAFT_REPO_ROOT/tools/bench_frames.py --batch 256 --pinned --reps 3 > /dev/null 2>&1
  echo "R=$r"; python3 - <<'PY'
import csv,glob
for f in glob.glob('/tmp/edtprof/**/*kernel_trace.csv', recursive=True):
    acc={}
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name'].split('(')[0].replace('void ','')
        if 'edt_' in n:
            acc.setdefault(n,[]).append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
    for n,v in acc.items(): print('  %-45s max %.1f us  n=%d'%(n,max(v)/1e3,len(v)))
PY
done
