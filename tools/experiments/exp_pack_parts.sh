#!/bin/bash
# where the rank-pack pass's time goes (results wrong, timing only): edtx6 no d2 loads, edtx7 no rank look-up, edtx8 no stores
cd /tmp; export TMPDIR=/tmp
for v in "" _edtx6 _edtx7 _edtx8; do
  rm -rf /tmp/edtprof; DVO_LIB_VARIANT=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/edtprof -o t -- python3 $GRAFT_REPO_ROOT/tools/bench_frames.py --batch 256 --pinned --reps 3 > /dev/null 2>&1
  echo "variant '$v'"; python3 - <<'PY'
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
